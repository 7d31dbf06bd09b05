// tests/emu/hip/hip_runtime.h — a SIMT emulator that stands in for <hip/hip_runtime.h> when the UNMODIFIED product sources
// (pg_embedding_amd/csrc/hnsw_gpu.hip and its device headers) are compiled for the host by tests/emu/build_emu.py.
//
// TEST INFRASTRUCTURE ONLY.  The product never builds, links or loads this: libhnsw_gpu.so is hipcc output for gfx950 and
// fails loudly without a device.  What this gives the CPU test tier is the kernels' own source executed with wavefront
// semantics, so that (a) the kernels' logic is compared with the oracle without a GPU and (b) the lock-free protocols
// between the waves of a block (team form: helpers, packages, jobs) run under real, preemptive thread schedules.
//
// Execution model
//   * a launch runs its blocks one after the other (no kernel here waits for another block);
//   * every wavefront of the running block is an OS thread: waves really race on LDS and global memory;
//   * the 64 lanes of a wave are coroutines (ucontext) of that thread.  A lane runs until it reaches a cross-lane operation
//     (ballot, readlane, readfirstlane, DPP, shuffle, wave barrier, __syncthreads), parks its operand and yields; when every
//     lane of the wave is parked (or has left the kernel) the operands are exchanged and the lanes continue.  Lanes that have
//     left the kernel count as inactive.  All parked lanes must be at the same KIND of operation — the product's kernels keep
//     cross-lane operations in wave-uniform control flow — otherwise the emulator aborts with a message;
//   * LDS is one array per block; atomics are real atomics; fences are full fences; s_sleep yields the processor.
// Not modelled: timing, the memory model's weakness (x86 is stronger), MFMA (the exhaustive scorer aborts here).
#pragma once
#define PGEMB_SIMT_EMULATOR 1      /* (launches are synchronous here: what needs a kernel that runs WHILE the host works says so) */
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <ucontext.h>
#include <x86intrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// ---- qualifiers --------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define HIP_KERNEL_NAME(...) __VA_ARGS__

// ---- vector types ------------------------------------------------------------------------------------------------------
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3
{
	uint32_t x, y, z;
	dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- the engine ----------------------------------------------------------------------------------------------------------
namespace simt {

enum Kind { K_NONE = 0, K_BALLOT, K_READLANE, K_READFIRST, K_DPP, K_SHFL, K_WAVE_BARRIER, K_BLOCK_BARRIER };

struct Block
{
	uint32_t bx = 0, nblocks = 0, nthreads = 0;
	pthread_barrier_t bar;
};

struct Wave
{
	static constexpr size_t STACK = 512 * 1024;
	uint32_t wib = 0;
	Block *blk = nullptr;
	const std::function<void()> *body = nullptr;
	ucontext_t sched;
	ucontext_t uc[64];
	char *stacks = nullptr;
	bool done[64], parked[64];
	int kind[64];
	uint64_t xchg[64], gather[64];
	uint64_t gmask = 0;                 // lanes that took part in the last exchange
	int cur = 0;
	uint32_t jitter = 0, jitter_us = 400;   // SIMT_EMU_JITTER=n: after one rendezvous in n the wave sleeps up to SIMT_EMU_JITTER_US (400) us
	uint64_t rng = 1;
};

inline thread_local Wave *tw = nullptr;

[[noreturn]] inline void die(const char *msg)
{
	fprintf(stderr, "simt emulator: %s\n", msg);
	abort();
}

inline void lane_entry()
{
	Wave *w = tw;
	(*w->body)();
	w = tw;
	w->done[w->cur] = true;             // uc_link takes the lane back to the scheduler
}

// park the calling lane with its operand; returns when every lane of the wave has arrived
inline void exchange(int kind, uint64_t v)
{
	Wave *w = tw;
	const int l = w->cur;
	w->kind[l] = kind;
	w->xchg[l] = v;
	w->parked[l] = true;
	swapcontext(&w->uc[l], &w->sched);
}

inline void run_wave(Wave *w)
{
	tw = w;
	for (int l = 0; l < 64; l++)
	{
		w->done[l] = false; w->parked[l] = false; w->kind[l] = K_NONE;
		getcontext(&w->uc[l]);
		w->uc[l].uc_stack.ss_sp = w->stacks + (size_t) l * Wave::STACK;
		w->uc[l].uc_stack.ss_size = Wave::STACK;
		w->uc[l].uc_link = &w->sched;
		makecontext(&w->uc[l], (void (*)()) lane_entry, 0);
	}
	for (;;)
	{
		for (int l = 0; l < 64; l++)
			if (!w->done[l]) { w->cur = l; swapcontext(&w->sched, &w->uc[l]); }
		uint64_t m = 0;
		int kind = K_NONE;
		for (int l = 0; l < 64; l++)
			if (!w->done[l])
			{
				if (!w->parked[l]) die("a lane came back to the scheduler without parking");
				if (kind == K_NONE) kind = w->kind[l];
				else if (kind != w->kind[l]) die("lanes of one wave wait at different kinds of cross-lane operations (divergent collective)");
				m |= 1ull << l;
			}
		if (!m) break;                                      // every lane has left the kernel
		for (int l = 0; l < 64; l++) { w->gather[l] = w->xchg[l]; w->parked[l] = false; }
		w->gmask = m;
		if (kind == K_BLOCK_BARRIER) pthread_barrier_wait(&w->blk->bar);
		else if (w->jitter)                                 // schedule fuzzing: now and then a wave falls asleep between two steps
		{
			w->rng ^= w->rng << 13; w->rng ^= w->rng >> 7; w->rng ^= w->rng << 17;
			if ((w->rng >> 11) % w->jitter == 0) std::this_thread::sleep_for(std::chrono::microseconds((w->rng >> 40) % w->jitter_us));
		}
	}
}

inline int lane_id() { return tw->cur; }

// ---- cross-lane operations (each one is ONE exchange) ---------------------------------------------------------------------
inline uint64_t ballot(bool p)
{
	exchange(K_BALLOT, p ? 1 : 0);
	const Wave *w = tw;
	uint64_t r = 0;
	for (int l = 0; l < 64; l++) if (((w->gmask >> l) & 1) && w->gather[l]) r |= 1ull << l;
	return r;
}
inline int readlane(int v, int lane)
{
	exchange(K_READLANE, (uint32_t) v);
	return (int) (uint32_t) tw->gather[lane & 63];
}
inline int readfirstlane(int v)
{
	exchange(K_READFIRST, (uint32_t) v);
	const Wave *w = tw;
	return (int) (uint32_t) w->gather[__builtin_ctzll(w->gmask)];
}
inline int shfl_xor(int v, int mask)
{
	exchange(K_SHFL, (uint32_t) v);
	return (int) (uint32_t) tw->gather[(lane_id() ^ mask) & 63];
}
inline float shfl_xor(float v, int mask)
{
	uint32_t u; memcpy(&u, &v, 4);
	u = (uint32_t) shfl_xor((int) u, mask);
	float r; memcpy(&r, &u, 4);
	return r;
}
// ds_bpermute_b32: every lane reads the operand of lane (addr / 4) mod 64 (0 from an inactive lane)
inline int ds_bpermute(int addr, int v)
{
	exchange(K_SHFL, (uint32_t) v);
	const Wave *w = tw;
	const int s = (addr >> 2) & 63;
	return ((w->gmask >> s) & 1) ? (int) (uint32_t) w->gather[s] : 0;
}
// ds_permute_b32: every lane WRITES its operand to lane (addr / 4) mod 64; a lane nobody writes to receives 0, and of several
// writers the highest lane wins (the kernels only ever use it with a permutation of the lanes)
inline int ds_permute(int addr, int v)
{
	exchange(K_SHFL, (uint32_t) ((addr >> 2) & 63));
	int dst[64];
	uint64_t act;
	{
		const Wave *w = tw;
		act = w->gmask;
		for (int l = 0; l < 64; l++) dst[l] = (int) (uint32_t) w->gather[l];
	}
	exchange(K_SHFL, (uint32_t) v);
	const Wave *w = tw;
	int r = 0;
	for (int l = 0; l < 64; l++)
		if (((act >> l) & 1) && dst[l] == w->cur) r = (int) (uint32_t) w->gather[l];
	return r;
}
inline void wave_barrier() { exchange(K_WAVE_BARRIER, 0); }
inline void syncthreads() { exchange(K_BLOCK_BARRIER, 0); }

// v_mov_b32_dpp with the controls the kernels use (quad_perm, row_mirror, row_half_mirror, row_bcast15/31, wave_shr:1);
// a lane whose row is masked out, or whose source lane does not exist or is inactive, keeps `old` (bound_ctrl = 0)
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	(void) bank_mask; (void) bound_ctrl;
	exchange(K_DPP, (uint32_t) src);
	const Wave *w = tw;
	const int i = w->cur, row = i >> 4;
	if (!((row_mask >> row) & 1)) return old;
	int s = -1;
	if (ctrl >= 0 && ctrl <= 0xFF) s = (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);
	else if (ctrl == 0x140) s = (i & ~15) | (15 - (i & 15));
	else if (ctrl == 0x141) s = (i & ~7) | (7 - (i & 7));
	else if (ctrl == 0x142) s = row >= 1 ? (row - 1) * 16 + 15 : -1;
	else if (ctrl == 0x143) s = row >= 2 ? 31 : -1;
	else if (ctrl == 0x138) s = i - 1;
	else if (ctrl == 0x130) s = i < 63 ? i + 1 : -1;
	else die("DPP control not modelled");
	if (s < 0 || !((w->gmask >> s) & 1)) return old;
	return (int) (uint32_t) w->gather[s];
}

inline uint32_t mbcnt_lo(uint32_t mask, uint32_t base)
{
	const int l = lane_id();
	return base + (uint32_t) __builtin_popcount(mask & (l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u)));
}
inline uint32_t mbcnt_hi(uint32_t mask, uint32_t base)
{
	const int l = lane_id();
	return base + (l > 32 ? (uint32_t) __builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}

// ---- launch -----------------------------------------------------------------------------------------------------------------
struct Idx { uint32_t tid = 0, bid = 0, bdim = 0, gdim = 0; };

// Blocks [b_from, b_to) of the grid, one after the other (a block = its wavefronts as threads).  poison = false: the caller has
// prepared the LDS image (resident launches: two block ranges run side by side over it, below).
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body, unsigned char *lds0, unsigned char *lds1, size_t lds_cap,
				   uint32_t b_from = 0, uint32_t b_to = 0xFFFFFFFFu, bool poison = true)
{
	if (block.x % 64 || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) die("launch shape not modelled (1-D, whole waves)");
	if (lds_bytes > lds_cap) die("more dynamic LDS than a CU has");
	const uint32_t nw = block.x / 64;
	static std::atomic<uint64_t> launches{0};
	const char *je = getenv("SIMT_EMU_JITTER"), *se = getenv("SIMT_EMU_SEED");
	const uint32_t jitter = je ? (uint32_t) atoi(je) : 0;
	const char *ue = getenv("SIMT_EMU_JITTER_US");
	const uint32_t jitter_us = ue && atoi(ue) > 0 ? (uint32_t) atoi(ue) : 400;
	const uint64_t seed = (se ? strtoull(se, nullptr, 10) : 0) * 1000003ull + launches.fetch_add(1);
	std::vector<Wave *> waves(nw);
	for (uint32_t i = 0; i < nw; i++)
	{
		waves[i] = new Wave();
		waves[i]->stacks = (char *) aligned_alloc(4096, 64 * Wave::STACK);
		if (!waves[i]->stacks) die("out of memory for lane stacks");
	}
	for (uint32_t b = b_from; b < grid.x && b < b_to; b++)
	{
		Block blk;
		blk.bx = b; blk.nblocks = grid.x; blk.nthreads = block.x;
		pthread_barrier_init(&blk.bar, nullptr, nw);
		if (poison)
		{
			memset(lds0, 0xA5, lds_bytes ? lds_bytes : 1);      // LDS is not zeroed on the device either
			memset(lds1, 0xA5, lds_bytes ? lds_bytes : 1);
		}
		std::vector<std::thread> th;
		for (uint32_t i = 0; i < nw; i++)
		{
			Wave *w = waves[i];
			w->wib = i; w->blk = &blk; w->body = &body;
			w->jitter = jitter; w->jitter_us = jitter_us; w->rng = (seed + b * 131 + i) * 0x9E3779B97F4A7C15ull + 1;
			th.emplace_back(run_wave, w);
		}
		for (auto &t : th) t.join();
		pthread_barrier_destroy(&blk.bar);
	}
	for (Wave *w : waves) { free(w->stacks); delete w; }
}

struct IdxX { operator uint32_t() const { return tw->wib * 64u + (uint32_t) tw->cur; } };
struct BidX { operator uint32_t() const { return tw->blk->bx; } };
struct BdimX { operator uint32_t() const { return tw->blk->nthreads; } };
struct GdimX { operator uint32_t() const { return tw->blk->nblocks; } };
struct One { operator uint32_t() const { return 1; } };
struct Zero { operator uint32_t() const { return 0; } };

}  // namespace simt

static const struct { simt::IdxX x; simt::Zero y, z; } threadIdx = {};
static const struct { simt::BidX x; simt::Zero y, z; } blockIdx = {};
static const struct { simt::BdimX x; simt::One y, z; } blockDim = {};
static const struct { simt::GdimX x; simt::One y, z; } gridDim = {};

// the block's LDS: kernels declare `extern __shared__ unsigned char smem[]` inside namespace pgemb or at global scope
constexpr size_t SIMT_LDS_BYTES = 160 * 1024;
alignas(16) static unsigned char smem[SIMT_LDS_BYTES];
namespace pgemb { alignas(16) static unsigned char smem[SIMT_LDS_BYTES]; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
	simt::launch_on((stream), dim3(grid), dim3(block), (size_t) (lds), [=]() { kernel(__VA_ARGS__); }, ::smem, pgemb::smem, SIMT_LDS_BYTES)

// ---- intrinsics ------------------------------------------------------------------------------------------------------------
#define __ballot(p) simt::ballot((bool) (p))
#define __shfl_xor(v, m) simt::shfl_xor((v), (m))
#define __syncthreads() simt::syncthreads()
#define __builtin_amdgcn_readlane(v, l) simt::readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) simt::readfirstlane((int) (v))
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) simt::update_dpp((o), (s), (c), (rm), (bm), (bc))
#define __builtin_amdgcn_wave_barrier() simt::wave_barrier()
#define __builtin_amdgcn_ds_bpermute(a, v) simt::ds_bpermute((a), (v))
#define __builtin_amdgcn_ds_permute(a, v) simt::ds_permute((a), (v))
#define __builtin_amdgcn_mbcnt_lo(m, b) simt::mbcnt_lo((m), (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) simt::mbcnt_hi((m), (b))
// A fence is ONE instruction of the wave: what any lane stored before it is before it for every lane.  Lanes are not in lockstep
// here, so a system-scope release — the one in front of a completion flag that the HOST polls while the launch runs (streams) — is also
// a rendezvous: without it lane 0 could raise the flag while another lane's result store is still to come (found by the stream
// scenario: the device's lockstep makes that impossible there).  All uses are in converged code.
namespace simt { inline void fence(int order, const char *scope) { __atomic_thread_fence(__ATOMIC_SEQ_CST); if (order == __ATOMIC_RELEASE && scope[0] == 0) wave_barrier(); } }
#define __builtin_amdgcn_fence(order, scope) simt::fence((order), (scope))
// s_waitcnt: every outstanding memory operation of the WAVE (all lanes) has completed — lanes are not in lockstep here, so it
// is a rendezvous (the kernels hand values from lane to lane through memory across it, e.g. beam_compact's scratch line)
#define __builtin_amdgcn_s_waitcnt(x) simt::wave_barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
#define __builtin_amdgcn_s_sleep(x) sched_yield()
#define __builtin_amdgcn_s_memtime() ((uint64_t) __rdtsc())
#define __builtin_amdgcn_s_memrealtime() ((uint64_t) (__rdtsc() >> 5))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) (simt::die("MFMA is not modelled"), (c))
// the exhaustive scorer's tile loads (global -> LDS directly) belong to the same unmodelled kernel
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) simt::die("global_load_lds is not modelled")
#define wall_clock64() ((unsigned long long) (__rdtsc() >> 5))

#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)

template <typename T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicCAS(T *p, T expect, T v)
{
	__atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
	return expect;
}
static inline uint32_t atomicAdd(uint32_t *p, int v) { return __atomic_fetch_add(p, (uint32_t) v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicExch(uint32_t *p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }

static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(uint32_t v) { float f; memcpy(&f, &v, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t v; memcpy(&v, &f, 4); return v; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t) (((uint64_t) a * b) >> 32); }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ---- runtime API: one "device", synchronous ----------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorInvalidValue = 1,
	   hipErrorInvalidResourceHandle = 400, hipErrorInvalidDevice = 101 };
// A stream runs its work at the call — except ONE kind of launch: a resident kernel that the host feeds while it runs (the library's
// streams, hnsw_gpu_stream_open, which says so through simt::next_launch_is_resident()).  That launch runs on threads of its own
// until it ends by itself; hipStreamQuery / hipEventQuery report hipErrorNotReady meanwhile, the ...Synchronize calls wait for it.
struct simt_stream { int dev = 0; std::thread *resident = nullptr; std::atomic<int> running{0}; };
typedef struct simt_stream *hipStream_t;
struct simt_event { std::chrono::steady_clock::time_point t; int dev = 0; simt_stream *after = nullptr; };
typedef simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 1, hipDeviceAttributeMultiprocessorCount = 2, hipDeviceAttributeWallClockRate = 3 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

// ---- several emulated devices (SIMT_EMU_DEVICES=N > 1) ------------------------------------------------------------------------
// What the host code of a multi-device path can get wrong is WHICH device something belongs to, and the emulator checks
// exactly that: every hipMalloc belongs to the device that was current (own pages); a kernel launched on device d runs with
// the memory of every other device inaccessible (PROT_NONE) unless d enabled peer access to it (SIMT_EMU_PEER=0: the
// "hardware" has no peer access at all) — touching it ends the process with a message; a launch or an event record on a
// stream of another device than the current one fails as HIP's does (hipErrorInvalidResourceHandle through hipGetLastError /
// the return value); hipMemcpyPeerAsync checks that both pointers belong to the devices it is told.  One kernel runs at a
// time (the protection is process-wide).  With one device none of this is active.
namespace simt {
struct DevAlloc { char *p; size_t bytes; int dev; };
inline std::mutex &rt_mu() { static std::mutex m; return m; }
inline std::vector<DevAlloc> &allocs() { static std::vector<DevAlloc> a; return a; }
inline thread_local int cur_dev = 0;
inline thread_local int last_error = 0;
inline int ndev() { static const int n = [] { const char *e = getenv("SIMT_EMU_DEVICES"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 16 ? v : 1; }(); return n; }
inline bool peer_hw() { static const bool p = [] { const char *e = getenv("SIMT_EMU_PEER"); return !(e && atoi(e) == 0); }(); return p; }
inline bool (&peer_on())[16][16] { static bool t[16][16] = {}; return t; }
inline std::atomic<int> &running_dev() { static std::atomic<int> d{-1}; return d; }
inline int owner_of(const void *q)
{
	for (const DevAlloc &a : allocs()) if ((const char *) q >= a.p && (const char *) q < a.p + a.bytes) return a.dev;
	return -1;
}
inline void segv_handler(int, siginfo_t *si, void *)
{
	char msg[256];
	int own = -1;
	for (const DevAlloc &a : allocs()) if ((char *) si->si_addr >= a.p && (char *) si->si_addr < a.p + a.bytes) own = a.dev;
	const int n = snprintf(msg, sizeof(msg), "SIMT emulator: a kernel on device %d touched %p, memory of device %d, without peer access\n",
						   running_dev().load(), si->si_addr, own);
	if (n > 0) (void) !write(2, msg, (size_t) n);
	_exit(86);
}
inline void protect_others(int dev, int prot)
{
	for (const DevAlloc &a : allocs())
		if (a.dev != dev && !peer_on()[dev][a.dev]) mprotect(a.p, a.bytes, prot);
}
inline thread_local bool resident_next = false;
inline void next_launch_is_resident() { resident_next = true; }
inline void wait_resident(simt_stream *s)
{
	if (s && s->resident)
	{
		s->resident->join();
		delete s->resident;
		s->resident = nullptr;
	}
}
inline void launch_on(simt_stream *stream, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body, unsigned char *lds0,
					  unsigned char *lds1, size_t lds_cap)
{
	if (resident_next)
	{
		// A resident launch: block 0 is its doorbell and runs for as long as the launch does, so it cannot be "the first of the blocks, one
		// after the other": it gets threads of its own next to the others (it never touches LDS: one wave that copies control words).  The
		// walking blocks still run one after the other — the first takes every query until the host stops the stream, the rest find the
		// stop word and leave: "as many blocks as the device holds at once" is one walking block here.
		resident_next = false;
		if (!stream || ndev() != 1 || grid.x < 2) die("a resident launch needs a stream, one device and a doorbell block + a walking block");
		wait_resident(stream);
		memset(lds0, 0xA5, lds_bytes ? lds_bytes : 1);
		memset(lds1, 0xA5, lds_bytes ? lds_bytes : 1);
		stream->running.store(1);
		const std::function<void()> kernel = body;              // (the caller's lambda dies with the call)
		stream->resident = new std::thread([=]() {
			std::thread doorbell([&]() { launch(grid, block, lds_bytes, kernel, lds0, lds1, lds_cap, 0, 1, false); });
			launch(grid, block, lds_bytes, kernel, lds0, lds1, lds_cap, 1, 0xFFFFFFFFu, false);
			doorbell.join();
			stream->running.store(0);
		});
		return;
	}
	if (ndev() == 1) { launch(grid, block, lds_bytes, body, lds0, lds1, lds_cap); return; }
	if (stream && stream->dev != cur_dev) { last_error = 400; return; }          // hipErrorInvalidResourceHandle: nothing runs
	std::lock_guard<std::mutex> g(rt_mu());
	static bool handler = false;
	if (!handler)
	{
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_sigaction = segv_handler; sa.sa_flags = SA_SIGINFO;
		sigaction(SIGSEGV, &sa, nullptr);
		handler = true;
	}
	running_dev().store(cur_dev);
	protect_others(cur_dev, PROT_NONE);
	launch(grid, block, lds_bytes, body, lds0, lds1, lds_cap);
	protect_others(cur_dev, PROT_READ | PROT_WRITE);
	running_dev().store(-1);
}
}  // namespace simt

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : (e == hipErrorOutOfMemory ? "out of memory" : "error"); }
static inline hipError_t hipGetLastError() { const int e = simt::last_error; simt::last_error = 0; return e; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = simt::ndev(); return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = simt::cur_dev; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= simt::ndev()) return hipErrorInvalidDevice; simt::cur_dev = d; return hipSuccess; }
static inline int simt_num_cu() { const char *e = getenv("SIMT_EMU_CUS"); const int n = e ? atoi(e) : 2; return n > 0 ? n : 2; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
	memset(p, 0, sizeof(*p));
	snprintf(p->name, sizeof(p->name), "SIMT emulator (tests)");
	snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950");
	p->multiProcessorCount = simt_num_cu();
	p->totalGlobalMem = (size_t) 64 << 30;
	return hipSuccess;
}
static inline hipError_t hipDeviceGetAttribute(int *v, int attr, int)
{
	*v = attr == hipDeviceAttributeMaxSharedMemoryPerBlock ? (int) SIMT_LDS_BYTES : attr == hipDeviceAttributeWallClockRate ? 100000 : simt_num_cu();
	return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// memory shared between processes: not modelled (one process is the whole "device" here)
struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipIpcMemLazyEnablePeerAccess = 1 };
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *, void *) { return hipErrorInvalidValue; }
static inline hipError_t hipIpcOpenMemHandle(void **, hipIpcMemHandle_t, unsigned) { return hipErrorInvalidValue; }
static inline hipError_t hipIpcCloseMemHandle(void *) { return hipErrorInvalidValue; }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer) { *can = simt::ndev() > 1 && simt::peer_hw() && dev != peer; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int peer, unsigned)
{
	if (peer < 0 || peer >= simt::ndev() || peer == simt::cur_dev || !simt::peer_hw()) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> g(simt::rt_mu());
	if (simt::peer_on()[simt::cur_dev][peer]) return hipErrorPeerAccessAlreadyEnabled;
	simt::peer_on()[simt::cur_dev][peer] = true;
	return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes)
{
	if (simt::ndev() > 1)                                   // own pages, owned by the current device
	{
		const size_t pg = 4096, len = ((bytes ? bytes : 1) + pg - 1) / pg * pg;
		void *q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (q == MAP_FAILED) return hipErrorOutOfMemory;
		std::lock_guard<std::mutex> g(simt::rt_mu());
		simt::allocs().push_back({(char *) q, len, simt::cur_dev});
		*p = (T *) q;
		return hipSuccess;
	}
	void *q = nullptr;
	if (posix_memalign(&q, 256, bytes ? bytes : 256)) return hipErrorOutOfMemory;
	*p = (T *) q;
	return hipSuccess;
}
static inline hipError_t hipFree(void *p)
{
	if (simt::ndev() > 1)
	{
		std::lock_guard<std::mutex> g(simt::rt_mu());
		auto &a = simt::allocs();
		for (size_t i = 0; i < a.size(); i++)
			if (a[i].p == (char *) p) { munmap(a[i].p, a[i].bytes); a.erase(a.begin() + (long) i); return hipSuccess; }
		return p ? hipErrorInvalidValue : hipSuccess;
	}
	free(p);
	return hipSuccess;
}
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return posix_memalign(p, 256, bytes ? bytes : 256) ? hipErrorOutOfMemory : hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int ddev, const void *s, int sdev, size_t n, hipStream_t = nullptr)
{
	if (simt::ndev() > 1)
	{
		std::lock_guard<std::mutex> g(simt::rt_mu());
		const int od = simt::owner_of(d), os = simt::owner_of(s);
		if ((od >= 0 && od != ddev) || (os >= 0 && os != sdev)) return hipErrorInvalidValue;    // a pointer that is not on the device it was said to be on
	}
	memcpy(d, s, n);
	return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = new simt_stream(); (*s)->dev = simt::cur_dev; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { simt::wait_resident(s); delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { simt::wait_resident(s); return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t s) { return s && s->running.load() ? hipErrorNotReady : hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new simt_event(); (*e)->dev = simt::cur_dev; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr)
{
	if (simt::ndev() > 1 && e->dev != (s ? s->dev : simt::cur_dev)) return hipErrorInvalidResourceHandle;   // event and stream of different devices
	e->t = std::chrono::steady_clock::now();
	e->after = s && s->running.load() ? s : nullptr;        // recorded behind a resident launch: reached when that launch has ended
	return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) { while (e->after && e->after->running.load()) sched_yield(); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t e) { return e->after && e->after->running.load() ? hipErrorNotReady : hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
	*ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
	return hipSuccess;
}
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <typename F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 1; return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind)
{
	for (size_t r = 0; r < height; r++) memcpy((char *) d + r * dpitch, (const char *) s + r * spitch, width);
	return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t = nullptr)
{
	return hipMemcpy2D(d, dpitch, s, spitch, width, height, k);
}
