// hnsw_gpu.hip — C-ABI implementation (include/hnsw_gpu.h) of the MI355X HNSW hot path.
// gfx950 only; plain HIP runtime, no framework types in any signature.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <chrono>
#include <new>
#include <vector>

#include "hnsw_gpu.h"
#include "search_kernels.h"
#include "device_build.h"
#include "device_insert.h"
#include "device_bf_mfma.h"
#include "device_roof.h"

using namespace pgemb;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIPCHK(expr)                                                                          \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return fail(HNSW_GPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
						__FILE__, __LINE__);                                                  \
	} while (0)

extern "C" const char *hnsw_gpu_last_error(void) { return g_err; }

extern "C" int hnsw_gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------
// configuration: resolved ONCE per process, never on a call path
// ------------------------------------------------------------------------------------
// Every knob of the library lives in one table of optional integers.  The table is filled from the environment at the first
// use (std::call_once) — the OPERATIONAL knobs only, the ones a deployment may want to set (include/hnsw_gpu.h,
// INTEGRATION.md "environment") — and after that a launch reads plain words: no getenv, no parsing, no locale on the path of
// a 0.4 ms call.  A host that wants another value later says so explicitly (hnsw_gpu_config_set; hnsw_gpu_config_reload re-reads
// the environment): that is how the test tiers flip kernel forms inside one process.  TEST knobs (forms and shapes the host code
// never picks by itself, forced so that every compiled kernel is exercised) can only be set through that call; experiment knobs
// of rejected variants exist only in -DHNSW_EXPERIMENT builds, where the whole table is read from the environment.
enum Knob : int
{
	// operational (environment, read once)
	K_BEAM, K_FORCE_LDS_HEAPS, K_TEAM, K_TEAM_MAX_NQ, K_WIDE_EF_MIN, K_REF_ORDER, K_NO_POLL, K_POLL_LIMIT_S, K_INSERT_FUSED,
	K_BLOCKS_PER_CU, K_STREAM_LIGHT,
	// test knobs (hnsw_gpu_config_set only)
	K_BEAM16, K_NARROW5, K_LEAN, K_HASH_ENTRIES, K_LDS_SET_MIN_WAVES, K_TEAM_SPEC, K_TEAM_WPB, K_MAX_BLOCKS, K_SHARDED_NO_PEER, K_BF_BIG_MIN_BLOCKS,
#ifdef HNSW_EXPERIMENT
	K_WIDE_WAVES, K_SHAPE_12X1, K_TEAM_MAINS, K_TEAM_COUNTERS,
#endif
	K_COUNT
};
struct KnobDef { const char *name; bool env; };
static const KnobDef g_knob_def[K_COUNT] = {
	{ "HNSW_GPU_BEAM", true }, { "HNSW_GPU_FORCE_LDS_HEAPS", true }, { "HNSW_GPU_TEAM", true }, { "HNSW_GPU_TEAM_MAX_NQ", true },
	{ "HNSW_GPU_WIDE_EF_MIN", true }, { "HNSW_GPU_REF_ORDER", true }, { "HNSW_GPU_NO_POLL", true }, { "HNSW_GPU_POLL_LIMIT_S", true },
	{ "HNSW_GPU_INSERT_FUSED", true }, { "HNSW_GPU_BLOCKS_PER_CU", true }, { "HNSW_GPU_STREAM_LIGHT", true },
	{ "HNSW_GPU_BEAM16", false }, { "HNSW_GPU_NARROW5", false }, { "HNSW_GPU_LEAN", false }, { "HNSW_GPU_HASH_ENTRIES", false }, { "HNSW_GPU_LDS_SET_MIN_WAVES", false },
	{ "HNSW_GPU_TEAM_SPEC", false }, { "HNSW_GPU_TEAM_WPB", false }, { "HNSW_GPU_MAX_BLOCKS", false }, { "HNSW_GPU_SHARDED_NO_PEER", false },
	{ "HNSW_GPU_BF_BIG_MIN_BLOCKS", false },
#ifdef HNSW_EXPERIMENT
	{ "HNSW_GPU_WIDE_WAVES", false }, { "HNSW_GPU_SHAPE_12X1", false }, { "HNSW_GPU_TEAM_MAINS", false }, { "HNSW_GPU_TEAM_COUNTERS", false },
#endif
};
struct KnobVal { std::atomic<long long> v{0}; std::atomic<bool> set{false}; };
static KnobVal g_knob[K_COUNT];
static std::once_flag g_knob_once;

static void knob_store(int k, const char *text)
{
	if (text && *text) { g_knob[k].v.store(atoll(text), std::memory_order_relaxed); g_knob[k].set.store(true, std::memory_order_release); }
	else g_knob[k].set.store(false, std::memory_order_release);
}

static void knobs_from_env(bool all)
{
	for (int k = 0; k < K_COUNT; k++)
	{
#ifdef HNSW_EXPERIMENT
		(void) all;
		knob_store(k, getenv(g_knob_def[k].name));
#else
		if (g_knob_def[k].env || all) knob_store(k, g_knob_def[k].env ? getenv(g_knob_def[k].name) : nullptr);
#endif
	}
}

static inline void knobs_init() { std::call_once(g_knob_once, [] { knobs_from_env(false); }); }

// value of knob k, or `dflt` when nobody set it
static inline long long knob(int k, long long dflt)
{
	return g_knob[k].set.load(std::memory_order_acquire) ? g_knob[k].v.load(std::memory_order_relaxed) : dflt;
}
static inline bool knob_is_set(int k) { return g_knob[k].set.load(std::memory_order_acquire); }

extern "C" int hnsw_gpu_config_set(const char *name, const char *value)
{
	if (!name) return fail(HNSW_GPU_ERR_ARG, "knob name is NULL");
	knobs_init();
	for (int k = 0; k < K_COUNT; k++)
		if (strcmp(name, g_knob_def[k].name) == 0) { knob_store(k, value); return HNSW_GPU_OK; }
	return fail(HNSW_GPU_ERR_ARG, "unknown knob %s", name);
}

extern "C" int hnsw_gpu_config_get(const char *name, long long *value)
{
	if (!name || !value) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	knobs_init();
	for (int k = 0; k < K_COUNT; k++)
		if (strcmp(name, g_knob_def[k].name) == 0)
		{
			if (!knob_is_set(k)) return 1;                  // known, at its default
			*value = knob(k, 0);
			return HNSW_GPU_OK;
		}
	return fail(HNSW_GPU_ERR_ARG, "unknown knob %s", name);
}

extern "C" void hnsw_gpu_config_reload(void)
{
	knobs_init();
	knobs_from_env(true);
}

// ------------------------------------------------------------------------------------
// the device mirror
// ------------------------------------------------------------------------------------
// Per-stream search state: the slots' visited bitmaps + logs, the ticket word and the HIP-event ring.
// Every mirror owns one (used by the plain entry points); hnsw_gpu_ctx adds more so that batches on
// different streams can be in flight at the same time.
struct SearchWs
{
	static const int EV_RING = 64;
	uint32_t *vis = nullptr;  size_t vis_slots = 0, vis_words = 0;
	uint32_t *vlog = nullptr; uint32_t logcap = 0;
	uint64_t *beam = nullptr; size_t beam_keys = 0;      // beam form: prune scratch, 64*UREG keys per slot
	uint64_t *sets = nullptr; size_t set_keys = 0;       // generic form with its sets in HBM: 3*ef+2 keys per slot
	uint32_t *ticket = nullptr;
	hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {};
	uint64_t launches = 0;
	uint32_t last_slots = 0;
	uint32_t walkers_hint = 0;                           // hnsw_gpu_ctx_set_walkers: walking waves per block of a small team launch (0 = by launch size)
	// stream mode, the next launch only (hnsw_gpu_stream_open): the host's control words, their device copies, ring size, walking waves per block
	const uint32_t *stream_host_next = nullptr; uint32_t *stream_dev_next = nullptr; uint32_t stream_ring_next = 0, stream_walkers_next = 0;
	uint32_t *done_next = nullptr;                       // completion flags for the next launch only
	uint32_t *pops_next = nullptr; uint32_t pops_cap_next = 0;   // pop-sequence output for the next launch only
	uint32_t *evals_next = nullptr; uint32_t evals_cap_next = 0; uint64_t *times_next = nullptr;   // evaluation trace, next launch only
	char kname[96] = "";                                 // symbol of the kernel the last launch used (as rocprofv3 prints it)
	uint32_t *team_dbg = nullptr;                        // 8 launch-wide counters of the team form (HNSW_GPU_TEAM_COUNTERS=1)
	// abort word (pinned host memory) + health counters (device memory): device_search.h, banner at abort_requested
	uint32_t *abort_host = nullptr;
	uint32_t *health = nullptr;
	int device = 0;
	int abort_sent = 0;                                  // (atomic) an abort was requested: the next launch re-zeroes the workspace
	uint32_t abort_requests = 0;                         // (atomic) abort requests this workspace has received in its life (hnsw_gpu_index_health [5])
	int64_t busy_since_ms = 0;                           // (atomic) steady-clock ms of the last launch, 0 = known idle (watchdog)
};

// ------------------------------------------------------------------------------------
// Abort + watchdog.  No wait inside the kernels is unbounded, so a launch that never ends would be a
// bug nobody has thought of; the abort word makes such a launch cost its caller's patience instead of
// the device: every wave reads the abort word at the top of a query / every 256 hops and leaves.
// Every workspace is registered here so that ANY thread (a test watchdog, a signal-safe helper thread,
// the library's own watchdog: HNSW_GPU_WATCHDOG_S=<seconds>) can reach it without the mirror's lock —
// the thread that owns the lock is the one that is stuck.
// ------------------------------------------------------------------------------------
// (never destroyed: the watchdog thread is detached and may still be looking at them while the process exits)
static std::mutex &g_ws_mu = *new std::mutex;
static std::vector<SearchWs *> &g_ws_all = *new std::vector<SearchWs *>;
static bool g_watchdog_started = false;

static int64_t now_ms()
{
	return (int64_t) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// g_ws_mu held.  The abort word lives in pinned host memory: a plain store, no HIP call, nothing that could queue behind
// the launch it is meant to end.
static int abort_ws_locked(SearchWs *w)
{
	if (!w->abort_host) return 0;
	__atomic_store_n(&w->abort_sent, 1, __ATOMIC_SEQ_CST);
	__atomic_add_fetch(&w->abort_requests, 1u, __ATOMIC_SEQ_CST);
	__atomic_store_n(w->abort_host, 1u, __ATOMIC_SEQ_CST);
	return 1;
}

extern "C" int hnsw_gpu_abort_all(void)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	int n = 0;
	for (SearchWs *w : g_ws_all) n += abort_ws_locked(w);
	return n;
}

static void watchdog_main(int limit_s)
{
	for (;;)
	{
		std::this_thread::sleep_for(std::chrono::milliseconds(500));
		std::lock_guard<std::mutex> g(g_ws_mu);
		const int64_t now = now_ms();
		for (SearchWs *w : g_ws_all)
		{
			const int64_t since = __atomic_load_n(&w->busy_since_ms, __ATOMIC_SEQ_CST);
			if (since == 0 || now - since < (int64_t) limit_s * 1000 || __atomic_load_n(&w->abort_sent, __ATOMIC_SEQ_CST)) continue;
			const uint64_t l = __atomic_load_n(&w->launches, __ATOMIC_SEQ_CST);
			if (l == 0) continue;
			hipEvent_t ev = w->ev1[(l - 1) % SearchWs::EV_RING];
			if (!ev || hipEventQuery(ev) != hipErrorNotReady)
			{
				// finished (or unknown): idle unless a newer launch has stamped it meanwhile
				int64_t expect = since;
				(void) __atomic_compare_exchange_n(&w->busy_since_ms, &expect, (int64_t) 0, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
				continue;
			}
			// The stamp was taken when the launch was ENQUEUED.  A deep queue of healthy asynchronous launches may keep it waiting
			// longer than the limit: while its start event has not completed it is not running, so the clock restarts now.
			hipEvent_t ev_start = w->ev0[(l - 1) % SearchWs::EV_RING];
			if (ev_start && hipEventQuery(ev_start) == hipErrorNotReady)
			{
				int64_t expect = since;
				(void) __atomic_compare_exchange_n(&w->busy_since_ms, &expect, now, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
				continue;
			}
			fprintf(stderr, "hnsw_gpu watchdog: search kernel %s on device %d has been running for %lld s (limit %d s): aborting it\n",
					w->kname, w->device, (long long) ((now - since) / 1000), limit_s);
			(void) abort_ws_locked(w);
		}
	}
}

// g_ws_mu NOT held
static void ws_register(SearchWs *w)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	g_ws_all.push_back(w);
	if (!g_watchdog_started)
	{
		g_watchdog_started = true;
		const char *e = getenv("HNSW_GPU_WATCHDOG_S");
		if (e && atoi(e) > 0) std::thread(watchdog_main, atoi(e)).detach();
	}
}

static void ws_unregister(SearchWs *w)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	g_ws_all.erase(std::remove(g_ws_all.begin(), g_ws_all.end(), w), g_ws_all.end());
}

static int ws_init(SearchWs *w)
{
	HIPCHK(hipMalloc(&w->ticket, 64));
	HIPCHK(hipMemset(w->ticket, 0, 64));
	HIPCHK(hipMalloc(&w->health, HEALTH_WORDS * 4));
	HIPCHK(hipMemset(w->health, 0, HEALTH_WORDS * 4));
	HIPCHK(hipHostMalloc((void **) &w->abort_host, 64, hipHostMallocDefault));
	memset(w->abort_host, 0, 64);
	(void) hipGetDevice(&w->device);
	for (int i = 0; i < SearchWs::EV_RING; i++)
	{
		HIPCHK(hipEventCreate(&w->ev0[i]));
		HIPCHK(hipEventCreate(&w->ev1[i]));
	}
	ws_register(w);
	return HNSW_GPU_OK;
}

static void ws_free(SearchWs *w)
{
	ws_unregister(w);
	if (w->health) (void) hipFree(w->health);
	if (w->abort_host) (void) hipHostFree(w->abort_host);
	if (w->vis) (void) hipFree(w->vis);
	if (w->beam) (void) hipFree(w->beam);
	if (w->sets) (void) hipFree(w->sets);
	if (w->vlog) (void) hipFree(w->vlog);
	if (w->team_dbg) (void) hipFree(w->team_dbg);
	if (w->ticket) (void) hipFree(w->ticket);
	for (int i = 0; i < SearchWs::EV_RING; i++)
	{
		if (w->ev0[i]) (void) hipEventDestroy(w->ev0[i]);
		if (w->ev1[i]) (void) hipEventDestroy(w->ev1[i]);
	}
	*w = SearchWs();
}

struct hnsw_gpu_index
{
	// One search / build / scratch user at a time per mirror: the public entry points that touch
	// the shared workspace take this lock (launches stay asynchronous on the caller's stream, but
	// two host threads must not interleave their launches on one handle).
	std::recursive_mutex mu;
	HnswMetadata meta;
	int      device = 0;
	int      num_cu = 0;
	size_t   max_lds = 64 * 1024;   // dynamic LDS one block may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
	bool     ins_dirty = false;     // an insert failed after its kernels were enqueued: block counters may be non-zero (insert_impl)
	size_t   n = 0, cap = 0;
	uint32_t stride = 0;      // floats per row (dim rounded up to 4)
	uint32_t lstride = 0;     // link slots per element (maxM rounded up to 16)
	float    *vec = nullptr;
	uint32_t *links = nullptr;
	uint64_t *labels = nullptr;
	SearchWs ws;              // default search state (grow-only)
	uint64_t generation = 0;  // bumped when capacity changes (bitmap width changes)
	uint32_t *misc = nullptr; // small device scratch words (import error counter, ...)
	// scratch for the host-pointer entry points
	void *scratch = nullptr; size_t scratch_bytes = 0;
	// pinned host staging of the few-queries host-pointer path (the kernel reads and writes it directly)
	char *pin = nullptr; size_t pin_bytes = 0;
	// hnsw_gpu_search_trace_begin .. _end
	bool trace_active = false; size_t trace_ef = 0, trace_cap = 0, trace_seen = 0; int trace_base = 0;
	// builder scratch (hnsw_gpu_index_link)
	void *bld = nullptr; size_t bld_batch = 0; size_t bld_tmp_bytes = 0;
	// single-insert scratch (device_insert.h): candidates of the insert's own walk | targets | pair matrix
	void *ins = nullptr; size_t ins_bytes = 0;
	// exhaustive MFMA scorer: |row|^2 cache + scratch
	float *xnorm = nullptr; size_t xnorm_n = 0, xnorm_cap = 0;
	void *bf = nullptr; size_t bf_bytes = 0;
	hipEvent_t bf_e0 = nullptr, bf_e1 = nullptr;
	// hnsw_gpu_search_batch, copy path: before the upload / after the last download (hnsw_gpu_last_batch_ms)
	hipEvent_t hb0 = nullptr, hb1 = nullptr; bool hb_valid = false;
};

static int ensure_scratch(hnsw_gpu_index *ix, size_t bytes)
{
	if (bytes <= ix->scratch_bytes) return HNSW_GPU_OK;
	if (ix->scratch) (void) hipFree(ix->scratch);
	ix->scratch = nullptr; ix->scratch_bytes = 0;
	HIPCHK(hipMalloc(&ix->scratch, bytes));
	ix->scratch_bytes = bytes;
	return HNSW_GPU_OK;
}

static int check_meta(const HnswMetadata *m)
{
	if (!m) return fail(HNSW_GPU_ERR_ARG, "meta is NULL");
	if (m->dim == 0 || m->dim > (1u << 20)) return fail(HNSW_GPU_ERR_ARG, "unsupported dim %zu", m->dim);
	if (m->maxM == 0 || m->maxM > 4096) return fail(HNSW_GPU_ERR_ARG, "unsupported maxM %zu", m->maxM);
	if ((int) m->dist_func < 0 || (int) m->dist_func > 2) return fail(HNSW_GPU_ERR_ARG, "bad dist_func %d", (int) m->dist_func);
	// element image offsets must be the ones of embedding.c:225-228
	if (m->offset_data != (m->maxM + 1) * sizeof(idx_t) || m->offset_label != m->offset_data + m->dim * sizeof(coord_t) ||
		m->size_data_per_element != m->offset_label + sizeof(label_t))
		return fail(HNSW_GPU_ERR_ARG, "meta offsets do not describe [count|links|vector|label]");
	return HNSW_GPU_OK;
}

static int alloc_index(const HnswMetadata *meta, size_t capacity, int device, hnsw_gpu_index **out)
{
	int rc = check_meta(meta);
	if (rc) return rc;
	int ndev = hnsw_gpu_device_count();
	if (ndev <= 0) return fail(HNSW_GPU_ERR_NODEVICE, "no HIP device visible (this library has no CPU path)");
	if (device < 0 || device >= ndev) return fail(HNSW_GPU_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
	if (capacity >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "capacity exceeds idx_t range");
	HIPCHK(hipSetDevice(device));
	hnsw_gpu_index *ix = new (std::nothrow) hnsw_gpu_index();
	if (!ix) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	ix->meta = *meta;
	ix->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) ix->num_cu = prop.multiProcessorCount;
	if (ix->num_cu <= 0) ix->num_cu = 256;
	{
		int maxlds = 0;
		if (hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && maxlds > 0) ix->max_lds = (size_t) maxlds;
	}
	ix->cap = capacity ? capacity : 1;
	ix->stride = (uint32_t) round_up(meta->dim, 4);
	ix->lstride = (uint32_t) round_up(meta->maxM, 16);
	hipError_t e;
	if ((e = hipMalloc(&ix->vec, ix->cap * ix->stride * sizeof(float))) != hipSuccess ||
		(e = hipMalloc(&ix->links, ix->cap * ix->lstride * sizeof(uint32_t))) != hipSuccess ||
		(e = hipMalloc(&ix->labels, ix->cap * sizeof(uint64_t))) != hipSuccess ||
		(e = hipMalloc(&ix->misc, 64)) != hipSuccess)
	{
		hnsw_gpu_index_destroy(ix);
		return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "index allocation failed: %s",
					hipGetErrorString(e));
	}
	(void) hipMemset(ix->misc, 0, 64);
	{
		int rc2 = ws_init(&ix->ws);
		if (rc2) { hnsw_gpu_index_destroy(ix); return rc2; }
	}
	*out = ix;
	return HNSW_GPU_OK;
}

extern "C" void hnsw_gpu_index_destroy(hnsw_gpu_index *ix)
{
	if (!ix) return;
	(void) hipSetDevice(ix->device);
	if (ix->vec) (void) hipFree(ix->vec);
	if (ix->links) (void) hipFree(ix->links);
	if (ix->labels) (void) hipFree(ix->labels);
	ws_free(&ix->ws);
	if (ix->misc) (void) hipFree(ix->misc);
	if (ix->scratch) (void) hipFree(ix->scratch);
	if (ix->pin) (void) hipHostFree(ix->pin);
	if (ix->bld) (void) hipFree(ix->bld);
	if (ix->ins) (void) hipFree(ix->ins);
	if (ix->xnorm) (void) hipFree(ix->xnorm);
	if (ix->bf) (void) hipFree(ix->bf);
	if (ix->bf_e0) (void) hipEventDestroy(ix->bf_e0);
	if (ix->bf_e1) (void) hipEventDestroy(ix->bf_e1);
	if (ix->hb0) (void) hipEventDestroy(ix->hb0);
	if (ix->hb1) (void) hipEventDestroy(ix->hb1);
	delete ix;
}

extern "C" size_t hnsw_gpu_index_count(const hnsw_gpu_index *ix) { return ix ? ix->n : 0; }
extern "C" size_t hnsw_gpu_index_capacity(const hnsw_gpu_index *ix) { return ix ? ix->cap : 0; }
extern "C" int    hnsw_gpu_index_device(const hnsw_gpu_index *ix) { return ix ? ix->device : -1; }

// ------------------------------------------------------------------------------------
// element image <-> mirror
// ------------------------------------------------------------------------------------

// One wavefront per element: [count|links|vector|label] -> links row / padded vector row / label.
// Everything in the image is 4-byte aligned only (embedding.c:226), so it is read as u32 words.
__global__ __launch_bounds__(256) void import_elements_kernel(const uint32_t *__restrict__ raw, size_t elem_words,
															  uint32_t first, uint32_t count, uint32_t n_total,
															  uint32_t dim, uint32_t stride, uint32_t maxM, uint32_t lstride,
															  float *vec, uint32_t *links, uint64_t *labels, uint32_t *bad)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	const uint32_t *src = raw + (size_t) w * elem_words;
	const size_t e = (size_t) first + w;
	uint32_t cnt = src[0];
	if (cnt > maxM) { if (lane == 0) atomicAdd(bad, 1u); cnt = maxM; }
	for (uint32_t j = lane; j < lstride; j += 64)
	{
		uint32_t t = LINK_NONE;
		if (j < cnt)
		{
			t = src[1 + j];
			if (t >= n_total) { atomicAdd(bad, 1u); t = LINK_NONE; }
			for (uint32_t k = 0; k < j && t != LINK_NONE; k++)      // keep the first occurrence only
				if (src[1 + k] == t) t = LINK_NONE;
		}
		links[e * lstride + j] = t;
	}
	const uint32_t *v = src + (maxM + 1);
	for (uint32_t c = lane; c < stride; c += 64)
		vec[e * stride + c] = (c < dim) ? __uint_as_float(v[c]) : 0.f;
	if (lane == 0)
	{
		const uint32_t *l = v + dim;
		labels[e] = (uint64_t) l[0] | ((uint64_t) l[1] << 32);
	}
}

__global__ __launch_bounds__(256) void export_elements_kernel(uint32_t *__restrict__ raw, size_t elem_words,
															  uint32_t first, uint32_t count,
															  uint32_t dim, uint32_t stride, uint32_t maxM, uint32_t lstride,
															  const float *vec, const uint32_t *links, const uint64_t *labels)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	uint32_t *dst = raw + (size_t) w * elem_words;
	const size_t e = (size_t) first + w;
	// links are stored compacted (no holes) with their count in front
	uint32_t cnt = 0;
	for (uint32_t j0 = 0; j0 < maxM; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		uint32_t t = (j < lstride && j < maxM) ? links[e * lstride + j] : LINK_NONE;
		const uint64_t m = __ballot(t != LINK_NONE);
		if (t != LINK_NONE) dst[1 + cnt + lane_rank(m)] = t;
		cnt += (uint32_t) __builtin_popcountll(m);
	}
	for (uint32_t j = cnt + lane; j < maxM; j += 64) dst[1 + j] = 0u;
	if (lane == 0) dst[0] = cnt;
	uint32_t *v = dst + (maxM + 1);
	for (uint32_t c = lane; c < dim; c += 64) v[c] = __float_as_uint(vec[e * stride + c]);
	if (lane == 0)
	{
		const uint64_t l = labels[e];
		v[dim] = (uint32_t) l;
		v[dim + 1] = (uint32_t) (l >> 32);
	}
}

// rows given as dim-strided floats (host order) -> padded rows, links cleared, labels set.
__global__ __launch_bounds__(256) void append_rows_kernel(const float *__restrict__ src, const uint64_t *__restrict__ src_labels,
														  uint32_t first, uint32_t count, uint32_t dim, uint32_t stride,
														  uint32_t lstride, float *vec, uint32_t *links, uint64_t *labels)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	const size_t e = (size_t) first + w;
	for (uint32_t c = lane; c < stride; c += 64)
		vec[e * stride + c] = (c < dim) ? src[(size_t) w * dim + c] : 0.f;
	for (uint32_t j = lane; j < lstride; j += 64) links[e * lstride + j] = LINK_NONE;
	if (lane == 0) labels[e] = src_labels ? src_labels[w] : (uint64_t) e;
}

static const size_t STAGE_BYTES = (size_t) 128 << 20;   // host<->device staging granule (two of them for uploads)

// Re-import `count` host element images into element numbers [first, first+count); the mirror
// grows to cover them.  n_total bounds the link targets that are accepted.  Two staging buffers, a copy
// stream and a kernel stream: the copy of granule i+1 runs while granule i is scattered into the mirror, and
// nothing waits for the whole device (other mirrors' searches keep running).
static int import_range(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count, size_t n_total)
{
	const HnswMetadata *meta = &ix->meta;
	const size_t esz = meta->size_data_per_element;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / esz);
	uint32_t *bad = ix->misc;
	if (count == 0) return HNSW_GPU_OK;
	HIPCHK(hipMemset(bad, 0, 4));
	const size_t nbuf = count > per ? 2 : 1;
	const size_t buf_bytes = round_up(std::min(per, count) * esz, 256);
	int rc0 = ensure_scratch(ix, nbuf * buf_bytes);                // staging lives in the mirror's scratch
	if (rc0) return rc0;
	hipStream_t s_copy = nullptr, s_kern = nullptr;
	hipEvent_t copied[2] = { nullptr, nullptr }, used[2] = { nullptr, nullptr };
	hipError_t e = hipSuccess;
	if (nbuf == 1)
	{
		// one granule (incremental updates, small mirrors): no pipeline to build — copy, scatter, and the read-back
		// of the error counter below waits for the kernel on the default stream
		e = hipMemcpy(ix->scratch, elements, count * esz, hipMemcpyHostToDevice);
		if (e == hipSuccess)
		{
			hipLaunchKernelGGL(import_elements_kernel, dim3((uint32_t) ((count + 3) / 4)), dim3(256), 0, 0, (const uint32_t *) ix->scratch,
							   esz / 4, (uint32_t) first, (uint32_t) count, (uint32_t) n_total, (uint32_t) meta->dim, ix->stride,
							   (uint32_t) meta->maxM, ix->lstride, ix->vec, ix->links, ix->labels, bad);
			e = hipGetLastError();
		}
		uint32_t nb1 = 0;
		if (e == hipSuccess) e = hipMemcpy(&nb1, bad, 4, hipMemcpyDeviceToHost);
		if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index upload failed: %s", hipGetErrorString(e));
		if (nb1) return fail(HNSW_GPU_ERR_ARG, "element image is corrupt: %u bad link counts / link targets", nb1);
		return HNSW_GPU_OK;
	}
	e = hipStreamCreate(&s_copy);
	if (e == hipSuccess) e = hipStreamCreate(&s_kern);
	for (int b = 0; b < 2 && e == hipSuccess; b++)
	{
		e = hipEventCreateWithFlags(&copied[b], hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&used[b], hipEventDisableTiming);
	}
	size_t chunk = 0;
	for (size_t off = 0; off < count && e == hipSuccess; off += per, chunk++)
	{
		const size_t cnt = std::min(per, count - off);
		const int b = (int) (chunk % nbuf);
		uint32_t *stage = (uint32_t *) ((char *) ix->scratch + (size_t) b * buf_bytes);
		if (chunk >= nbuf) e = hipStreamWaitEvent(s_copy, used[b], 0);       // the kernel that read this buffer is done
		if (e == hipSuccess) e = hipMemcpyAsync(stage, (const char *) elements + off * esz, cnt * esz, hipMemcpyHostToDevice, s_copy);
		if (e == hipSuccess) e = hipEventRecord(copied[b], s_copy);
		if (e == hipSuccess) e = hipStreamWaitEvent(s_kern, copied[b], 0);
		if (e != hipSuccess) break;
		const uint32_t blocks = (uint32_t) ((cnt + 3) / 4);
		hipLaunchKernelGGL(import_elements_kernel, dim3(blocks), dim3(256), 0, s_kern, stage, esz / 4, (uint32_t) (first + off),
						   (uint32_t) cnt, (uint32_t) n_total, (uint32_t) meta->dim, ix->stride, (uint32_t) meta->maxM,
						   ix->lstride, ix->vec, ix->links, ix->labels, bad);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipEventRecord(used[b], s_kern);
	}
	if (s_copy) { const hipError_t e2 = hipStreamSynchronize(s_copy); if (e == hipSuccess) e = e2; }
	if (s_kern) { const hipError_t e2 = hipStreamSynchronize(s_kern); if (e == hipSuccess) e = e2; }
	uint32_t nbad = 0;
	if (e == hipSuccess) e = hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
	for (int b = 0; b < 2; b++)
	{
		if (copied[b]) (void) hipEventDestroy(copied[b]);
		if (used[b]) (void) hipEventDestroy(used[b]);
	}
	if (s_copy) (void) hipStreamDestroy(s_copy);
	if (s_kern) (void) hipStreamDestroy(s_kern);
	if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index upload failed: %s", hipGetErrorString(e));
	if (nbad) return fail(HNSW_GPU_ERR_ARG, "element image is corrupt: %u bad link counts / link targets", nbad);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_create_from_flat(const HnswMetadata *meta, const void *elements, size_t n,
											   int device, hnsw_gpu_index **out)
{
	if (!out) return fail(HNSW_GPU_ERR_ARG, "out is NULL");
	if (n && !elements) return fail(HNSW_GPU_ERR_ARG, "elements is NULL");
	hnsw_gpu_index *ix = nullptr;
	int rc = alloc_index(meta, n, device, &ix);
	if (rc) return rc;
	rc = import_range(ix, elements, 0, n, n);
	if (rc) { hnsw_gpu_index_destroy(ix); return rc; }
	ix->n = n;
	*out = ix;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_create_empty(const HnswMetadata *meta, size_t capacity, int device, hnsw_gpu_index **out)
{
	if (!out) return fail(HNSW_GPU_ERR_ARG, "out is NULL");
	return alloc_index(meta, capacity, device, out);
}

extern "C" int hnsw_gpu_index_append_dev(hnsw_gpu_index *ix, const coord_t *d_vectors, const label_t *d_labels,
										 size_t n, void *stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (n == 0) return HNSW_GPU_OK;
	if (!d_vectors) return fail(HNSW_GPU_ERR_ARG, "vectors is NULL");
	if (ix->n + n > ix->cap) return fail(HNSW_GPU_ERR_ARG, "append exceeds capacity (%zu + %zu > %zu)", ix->n, n, ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t blocks = (uint32_t) ((n + 3) / 4);
	hipLaunchKernelGGL(append_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, d_vectors, d_labels,
					   (uint32_t) ix->n, (uint32_t) n, (uint32_t) ix->meta.dim, ix->stride, ix->lstride, ix->vec,
					   ix->links, ix->labels);
	HIPCHK(hipGetLastError());
	ix->n += n;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_append(hnsw_gpu_index *ix, const coord_t *vectors, const label_t *labels, size_t n)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (n == 0) return HNSW_GPU_OK;
	if (!vectors) return fail(HNSW_GPU_ERR_ARG, "vectors is NULL");
	if (ix->n + n > ix->cap) return fail(HNSW_GPU_ERR_ARG, "append exceeds capacity (%zu + %zu > %zu)", ix->n, n, ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	const size_t dim = ix->meta.dim;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / (dim * 4 + 8));
	int rc = ensure_scratch(ix, std::min(per, n) * (dim * 4 + 8) + 8);      // + alignment slack for the label array
	if (rc) return rc;
	for (size_t first = 0; first < n; first += per)
	{
		const size_t cnt = std::min(per, n - first);
		float *dv = (float *) ix->scratch;
		uint64_t *dl = (uint64_t *) ((char *) ix->scratch + round_up(cnt * dim * 4, 8));
		HIPCHK(hipMemcpy(dv, vectors + first * dim, cnt * dim * 4, hipMemcpyHostToDevice));
		if (labels) HIPCHK(hipMemcpy(dl, labels + first, cnt * 8, hipMemcpyHostToDevice));
		rc = hnsw_gpu_index_append_dev(ix, dv, labels ? dl : nullptr, cnt, nullptr);
		if (rc) return rc;
		HIPCHK(hipDeviceSynchronize());
	}
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_export_flat(hnsw_gpu_index *ix, void *elements)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !elements) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(ix->device));
	const size_t esz = ix->meta.size_data_per_element;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / esz);
	uint32_t *stage = nullptr;
	HIPCHK(hipMalloc(&stage, std::min(per, std::max<size_t>(ix->n, 1)) * esz));
	hipError_t e = hipSuccess;
	for (size_t first = 0; first < ix->n && e == hipSuccess; first += per)
	{
		const size_t cnt = std::min(per, ix->n - first);
		const uint32_t blocks = (uint32_t) ((cnt + 3) / 4);
		hipLaunchKernelGGL(export_elements_kernel, dim3(blocks), dim3(256), 0, 0, stage, esz / 4, (uint32_t) first,
						   (uint32_t) cnt, (uint32_t) ix->meta.dim, ix->stride, (uint32_t) ix->meta.maxM, ix->lstride,
						   ix->vec, ix->links, ix->labels);
		e = hipMemcpy((char *) elements + first * esz, stage, cnt * esz, hipMemcpyDeviceToHost);
	}
	(void) hipFree(stage);
	if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index download failed: %s", hipGetErrorString(e));
	return HNSW_GPU_OK;
}

// vacuum flags of a batch of elements: one upload of the element numbers + one launch (a VACUUM flags many rows,
// embedding.c:883-946; doing them one blocking copy pair at a time is a synchronisation storm)
__global__ __launch_bounds__(256) void set_deleted_kernel(uint64_t *__restrict__ labels, const uint32_t *__restrict__ idx,
														 size_t count, int deleted)
{
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	const unsigned long long bit = 1ull << HNSW_LABEL_DELETED_BIT;
	unsigned long long *l = reinterpret_cast<unsigned long long *>(labels + idx[i]);
	if (deleted) atomicOr(l, bit); else atomicAnd(l, ~bit);        // atomic: the same element may be listed twice
}

extern "C" int hnsw_gpu_index_set_deleted_batch(hnsw_gpu_index *ix, const idx_t *idx, size_t count, int deleted)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (count == 0) return HNSW_GPU_OK;
	if (!idx) return fail(HNSW_GPU_ERR_ARG, "NULL element list");
	for (size_t i = 0; i < count; i++)
		if (idx[i] >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx[i]);
	HIPCHK(hipSetDevice(ix->device));
	int rc = ensure_scratch(ix, count * sizeof(uint32_t));
	if (rc) return rc;
	HIPCHK(hipMemcpy(ix->scratch, idx, count * sizeof(uint32_t), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(set_deleted_kernel, dim3((uint32_t) ((count + 255) / 256)), dim3(256), 0, 0,
					   ix->labels, (const uint32_t *) ix->scratch, count, deleted);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(0));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_set_deleted(hnsw_gpu_index *ix, idx_t idx, int deleted)
{
	return hnsw_gpu_index_set_deleted_batch(ix, &idx, 1, deleted);
}

// ------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------
// which kernel a launch runs: search_kernels.h (one translation unit per load shape)
static search_kernel_t pick_search_kernel(int func, uint32_t kiters, int rreg, bool team, bool narrow5, bool lean)
{
	switch (shape_index(kiters))
	{
		case 0:
			if (narrow5 && !team) return pick_kernel_shape2x2(func, rreg, lean);      // the hot narrow-row form: 8 rows per pass, 96 VGPRs, 5 waves/SIMD
			return pick_kernel_shape2x4(func, rreg, team);
		case 1:  return pick_kernel_shape4x2(func, rreg, team);
		case 2:  return pick_kernel_shape8x2(func, rreg, team);
		default:
#ifdef HNSW_EXPERIMENT
			if (knob(K_SHAPE_12X1, 0)) return pick_kernel_shape12x1(func, rreg, team);
#endif
			return pick_kernel_shape12x2(func, rreg, team);
	}
}

static const size_t LDS_PER_CU = 160 * 1024;
static const size_t VIS_BUDGET_BYTES = (size_t) 24 << 30;     // cap on bitmap workspace
static const size_t SET_BUDGET_BYTES = (size_t) 8 << 30;      // cap on the HBM result/candidate areas (generic form)
// (no cap on the effective beam: beyond WIDE_EF_MIN the wide-beam form keeps both sets with a second level of chunk extremes,
// device_search_wide.h; what bounds a beam is the per-slot scratch, 24 bytes per result slot, under SET_BUDGET_BYTES)
static const size_t WIDE_EF_MIN = 2048;

static int launch_search(hnsw_gpu_index *ix, SearchWs *w, const float *d_queries, size_t q_stride, size_t nq, size_t ef, int mode,
						 uint64_t *d_labels, uint32_t *d_idx, float *d_dists, uint32_t *d_counts,
						 uint32_t *d_stats, hipStream_t stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_queries || !d_counts || (mode == 0 && !d_labels) || (mode == 1 && !d_idx))
		return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	if (nq >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "too many queries");
	// A beam wider than the index behaves exactly like a beam of the index size (nothing is ever evicted,
	// the walk ends when the candidates run out), so the scan's efSearch doubling (embedding.c:334) can go
	// as far as it likes; the output arrays keep the caller's ef as their row stride.
	const size_t out_stride = ef;
	ef = std::min(ef, std::max<size_t>(ix->n, 1));
	HIPCHK(hipSetDevice(ix->device));

	SearchArgs a;
	memset(&a, 0, sizeof(a));
	a.vec = ix->vec; a.links = ix->links; a.labels = ix->labels;
	a.n = (uint32_t) ix->n; a.dim = (uint32_t) ix->meta.dim; a.stride = ix->stride;
	a.nchunks = ix->stride / 4; a.kiters = (a.nchunks + 15) / 16;
	a.maxM = (uint32_t) ix->meta.maxM; a.lstride = ix->lstride; a.entry = ix->meta.enterpoint_node;
	a.queries = d_queries; a.q_stride = (uint32_t) q_stride; a.nq = (uint32_t) nq; a.ef = (uint32_t) ef; a.ccap = (uint32_t) (2 * ef);
	a.out_stride = (uint32_t) out_stride;
	a.out_labels = d_labels; a.out_idx = d_idx; a.out_dists = d_dists; a.out_counts = d_counts; a.out_stats = d_stats;
	a.mode = mode;
	if (a.n > 0 && a.entry >= a.n) return fail(HNSW_GPU_ERR_ARG, "enterpoint_node %u >= count %u", a.entry, a.n);

	// LDS carve per wave
	a.qpad_floats = (uint32_t) round_up(a.kiters, shape_kb(shape_index(a.kiters))) * 64;
	// Form of the accepted-set bookkeeping (rreg):
	//   beam form (one accepted set in registers, acceptance by counting): default up to ef = 256, and up
	//     to ef = 512 for rows wider than 256 floats — those run at 2 waves/SIMD anyway, so 16 set registers
	//     beat the LDS form there (+22-28 %, profiles/r1i_beam_form.txt); narrow rows keep the LDS form
	//     above 256 (it holds 4 waves/SIMD).  Its prune packs the "expanded" bit into bit 31 of the idx.
	//   LDS (generic) form: everything else — also HNSW_GPU_BEAM=0 and mirrors of >= 2^31 elements (HNSW_GPU_FORCE_LDS_HEAPS=1 forces it).
	//   (two-set register form, round 1's hot kernel: experiment builds only since round 5.)
	// The register forms use their LDS "res"/"cand" areas only as scratch of the emit step.
	knobs_init();
	const bool use_beam = knob(K_BEAM, 1) != 0 && ix->cap < 0x80000000ull;
	const bool beam16 = use_beam && ef > 256 && ef <= 512 && (knob_is_set(K_BEAM16) ? knob(K_BEAM16, 0) > 0 : shape_index(a.kiters) >= 2);
	const size_t wide_min = (size_t) knob(K_WIDE_EF_MIN, (long long) WIDE_EF_MIN);
	// Debug arithmetic (device_dist.h, F_L2_REF / F_MANHATTAN_REF / F_COSINE_REF): the summation order of oracle/_ref's own build, for a query-by-query
	// comparison of id lists with the compiled reference.  One kernel set only: beam form, 4 set registers, one wave per query.
	int func_code = (int) ix->meta.dist_func;
	bool reforder = false;
	if (knob(K_REF_ORDER, 0) > 0)
		{
			const bool ok = ef <= 128 && ix->cap < 0x80000000ull &&
							((func_code == F_L2 && ix->meta.dim % 16 == 0) || ((func_code == F_MANHATTAN || func_code == F_COSINE) && ix->meta.dim % 4 == 0));
			if (!ok)
				return fail(HNSW_GPU_ERR_ARG, "HNSW_GPU_REF_ORDER: only L2 with dims %% 16 == 0 or cosine / Manhattan with dims %% 4 == 0, ef <= 128");
			reforder = true;
			func_code = func_code == F_L2 ? F_L2_REF : (func_code == F_COSINE ? F_COSINE_REF : F_MANHATTAN_REF);
		}
	int rreg;
	if (reforder) rreg = -4;
	else if (ef > wide_min) rreg = 3;
	else if (knob(K_FORCE_LDS_HEAPS, 0) > 0) rreg = 0;
	else if (use_beam && (ef <= 256 || beam16)) rreg = ef <= 64 ? -2 : (ef <= 128 ? -4 : (ef <= 256 ? -8 : -16));
#ifdef HNSW_EXPERIMENT
	else rreg = ef <= 128 ? 2 : (ef <= 256 ? 4 : 0);         // two-set register form (experiment builds)
#else
	else rreg = 0;                                           // HNSW_GPU_BEAM=0, or a mirror of >= 2^31 elements: the generic form
#endif
	const size_t ucap = rreg < 0 ? (size_t) 64 * (size_t) -rreg : 0;      // beam form: slots of the accepted set
	// Team form wanted for this launch?  (decided for good further down, once the LDS carve is known)
	const int treq = (int) knob(K_TEAM, -1);
	const size_t auto_nq = (size_t) knob(K_TEAM_MAX_NQ, (long long) ix->num_cu);
	const bool stream_launch = w->stream_host_next != nullptr;
	const bool team_wanted = rreg < 0 && !reforder && (stream_launch || (treq != 0 && (treq > 0 || ix->stride > 320 || nq <= auto_nq)));
	// narrow rows, hot form: beam kernel with <= 4 set registers, one sum per row (L2 / Manhattan), not a team
	const bool narrow5 = shape_index(a.kiters) == 0 && (rreg == -2 || rreg == -4) && (int) ix->meta.dist_func != F_COSINE &&
						 !team_wanted && !reforder && knob(K_NARROW5, 1) != 0;
	size_t off = (size_t) a.qpad_floats * 4;
	if (rreg == 3)
	{
		// wide-beam form: both sets in the slot's HBM area [res: P | cand: 2P], P = the power of two >= ef (the output sort is a
		// bitonic network); per-chunk extremes in LDS: chunks of >= 1024 keys, at most 1024 chunks of candidates
		size_t P = 2;
		while (P < ef) P <<= 1;
		size_t ch = 1024;
		while ((2 * ef + ch - 1) / ch > 1024) ch <<= 1;
		a.wide_p = (uint32_t) P; a.wide_ch = (uint32_t) ch;
		a.wide_nr = (uint32_t) ((ef + ch - 1) / ch); a.wide_nc = (uint32_t) ((2 * ef + ch - 1) / ch);
		a.set_stride = 3 * P;
		a.off_res = (uint32_t) off;  off += round_up((size_t) a.wide_nr * 8, 16);
		a.off_cand = (uint32_t) off; off += round_up((size_t) a.wide_nc * 8, 16);
		if (3 * P * 8 > SET_BUDGET_BYTES)
			return fail(HNSW_GPU_ERR_NOMEM, "ef %zu needs %zu bytes of scratch per query slot (more than the %zu-byte budget)", ef, 3 * P * 8, SET_BUDGET_BYTES);
	}
	else if (rreg)
	{
		// [query | hash set (overlaid by the emit step's tie scratch) | newid | newdist]
		const size_t fixed = off + 64 * 4 + 128 * 4 + (team_wanted ? sizeof(TeamCtl) : 0);   // (+ the wave's control block behind the regions)
		// Rows of >= 1.25 KiB make the traversal HBM-bound, and there the LDS set pays (no L2
		// atomics, ~10 % less HBM traffic; measured 5.4 -> 7.5 TB/s at 768 dims) at 8 waves per CU.
		// Narrow rows are latency-bound and want 16-20 waves per CU: the beam form gives them a bucketed set of
		// 3456-4096 16-bit tags (ids whose bucket is full go to the bitmap); the two-set register form keeps the
		// bitmap only (profiles/r1g_visited_set_by_dim.txt, profiles/r1i_beam_form.md, profiles/r2m_*).
		// (rows of up to 128 floats in the beam form with ef <= 128, L2 / Manhattan, launches that will not run as
		// teams: 5 waves/SIMD with the 8-rows-per-pass shape — measured +6-10 % over 4 waves, profiles/r2m_*)
		const bool wide = ix->stride > 320;
		size_t want_waves = wide ? 8 : (narrow5 ? 20 : 16);
#ifdef HNSW_EXPERIMENT
		if (wide && knob(K_WIDE_WAVES, 0) >= 4) want_waves = (size_t) knob(K_WIDE_WAVES, 0);   // (experiment builds at 3 waves/SIMD)
#endif
		uint32_t hcap = wide ? 4096 : (rreg < 0 ? 2048 : 0);
		if (knob_is_set(K_HASH_ENTRIES)) hcap = (uint32_t) knob(K_HASH_ENTRIES, 0);
		// emit scratch: [keys | labels]; the beam form sorts up to `ucap` survivors (ties at the bound)
		const size_t nkeys = ucap ? ucap : ef;
		const size_t emit = round_up(nkeys * 8, 16) + round_up(ef * 8, 16);
		if (rreg < 0)
		{
			// beam form: hcap/4 buckets (any count, 128-byte steps of LDS) of eight 16-bit tags (device_search.h,
			// "bucketed"); tag = id / buckets + 1 must fit 16 bits and the 38-bit reciprocal must be exact (ids below
			// 2^28), else the kernel runs on the HBM bitmap alone
			hcap = std::min<uint32_t>(hcap, 4096);
			while (hcap >= 512 && want_waves * (fixed + std::max<size_t>(hcap * 4, emit)) > LDS_PER_CU) hcap -= 128;
			hcap &= ~31u;
			if (hcap < 512 || (uint64_t) ix->cap > (uint64_t) 65535 * (hcap / 4) || ix->cap >= (1u << 28)) hcap = 0;
			a.hmagic = hcap ? (uint32_t) ((((uint64_t) 1 << 38) + hcap / 4 - 1) / (hcap / 4)) : 0;
		}
		else
		{
			while (hcap >= 512 && want_waves * (fixed + std::max<size_t>(hcap * 4, emit)) > LDS_PER_CU) hcap >>= 1;
			if (hcap < 512 || (hcap & (hcap - 1))) hcap = 0;
		}
		a.hcap = hcap;
		a.hmax = hcap - hcap / 4;
		a.off_hash = (uint32_t) off;
		a.off_res = (uint32_t) off;
		a.off_cand = (uint32_t) (off + round_up(nkeys * 8, 16));
		off += round_up(std::max<size_t>((size_t) hcap * 4, emit), 16);
	}
	else
	{
		// generic form: [res ef+1 | cand 2ef+1] keys per wave — in LDS while at least HNSW_GPU_LDS_SET_MIN_WAVES
		// (default 4) waves per CU fit, otherwise in a per-slot HBM area (any ef)
		const size_t set_bytes = round_up((ef + 1) * 8, 16) + round_up((2 * ef + 1) * 8, 16);
		const size_t min_waves = knob(K_LDS_SET_MIN_WAVES, 0) > 0 ? (size_t) knob(K_LDS_SET_MIN_WAVES, 0) : 4;
		if (min_waves * (off + set_bytes + 64 * 4 + 128 * 4) > LDS_PER_CU)
		{
			rreg = 1;
			a.off_res = 0;
			a.off_cand = (uint32_t) (ef + 1);                     // in keys, inside the slot's area
			a.set_stride = 3 * ef + 2;
		}
		else
		{
			a.off_res = (uint32_t) off;     off += round_up((ef + 1) * 8, 16);
			a.off_cand = (uint32_t) off;    off += round_up((2 * ef + 1) * 8, 16);
		}
	}
	a.off_newid = (uint32_t) off;   off += 64 * 4;
	a.off_newdist = (uint32_t) off; off += 128 * 4;      // sums + (cosine) |x|^2
	a.wave_bytes = (uint32_t) round_up(off, 16);
	if (a.wave_bytes > LDS_PER_CU)
		return fail(HNSW_GPU_ERR_ARG, "ef=%zu dim=%zu needs %u bytes of LDS per query (> %zu)", ef, ix->meta.dim,
					a.wave_bytes, LDS_PER_CU);
	uint32_t wpb = 4;
	while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1;
	// Team form of the beam kernel (device_search.h, "Team form"): waves of a block that have no query (left) help a
	// sibling's walk with packages prepared in their own, otherwise idle LDS regions.  Measured at 1M rows
	// (profiles/r2_team_form.txt): rows wider than 320 floats gain at every launch size (one query 0.68 -> 0.47 ms,
	// 256 queries -24 %, 10 000 -3 %, 40 000 -0.7 %: only the tail of a big launch has idle waves); narrow rows gain
	// up to ~256 queries per launch and lose beyond (the larger kernel costs the 4-waves-per-SIMD steady state 10-16 %).
	// HNSW_GPU_TEAM=0/1 forces it off/on, HNSW_GPU_TEAM_MAX_NQ moves the narrow-row threshold, HNSW_GPU_TEAM_WPB the
	// waves per block (default 8 when the LDS of a block allows).
	bool team = false;
	if (rreg < 0)
	{
		const size_t pub = (size_t) 64 * ucap / 64 * 8;                 // 64*UREG keys
		// a donated region: [accepted-set copy | expanded bits | miss ids | package headers | packages | memo], all below
		// off_newid.  As many package slots as leave a useful memo: an element packaged while it was 6th in line may
		// be popped dozens of hops later, and a direct-mapped slot that was reused by then is a lost package.
		uint32_t lcs = 32;
		size_t o_ex = pub, o_miss = o_ex + 256, o_tag = round_up(o_miss + 256, 8), o_state = 0, o_links = 0, o_dc = 0, dccap = 0;
		for (; lcs >= 4; lcs >>= 1)
		{
			o_state = o_tag; o_links = o_tag + lcs * 8;                          // headers: lcs x u64; packages: lcs x lstride x u64
			o_dc = round_up(o_links + (size_t) lcs * a.lstride * 8, 16);
			const size_t want = lcs >= 16 ? 512 : (lcs == 8 ? 256 : 128);         // memo entries this many slots must leave
			dccap = 0;
			if (o_dc + want * 8 <= a.off_newid)
			{
				dccap = want;
				while (o_dc + dccap * 2 * 8 <= a.off_newid && dccap < 2048) dccap *= 2;
				break;
			}
		}
		if (dccap >= 128 && team_wanted)
		{
			team = true;
			a.tm_off_ex = (uint32_t) o_ex; a.tm_off_miss = (uint32_t) o_miss; a.tm_off_lctag = (uint32_t) o_tag;
			a.tm_off_lcstate = (uint32_t) o_state; a.tm_off_lclinks = (uint32_t) o_links; a.tm_lcslots = lcs;
			a.tm_off_dc = (uint32_t) o_dc; a.tm_dccap = (uint32_t) dccap;
			{
				// helpers of rank < tm_spec prepare packages ahead of the walk; the others score slices of its many-row hops
				// (device_search.h, banner at TeamCtl).  Measured at 1M rows (profiles/r3a_slice_helpers.txt): 768 dims, 5 of
				// 7 helpers speculating: one query 0.470 -> 0.438 ms, 16 queries -3.4 %, 256 -4.1 %, 1024 -3.4 %, 10 000 -0.5 %,
				// 40 000 -0.2 %; 3: 0.452; 0 (nobody speculates): 0.618.  128 dims: a hop rarely has more rows than one pass of 16,
				// slices lose 1-2 %, so narrow rows let every helper speculate.  HNSW_GPU_TEAM_SPEC overrides (8 = all speculate).
				a.tm_spec = knob_is_set(K_TEAM_SPEC) ? (uint32_t) std::max<long long>(0, knob(K_TEAM_SPEC, 0)) : (ix->stride > 320 ? 5u : 8u);
			}
			int maxlds = 64 * 1024;
			(void) hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, ix->device);
			uint32_t want = knob(K_TEAM_WPB, 0) > 0 ? (uint32_t) knob(K_TEAM_WPB, 0) : 8u;
			want = std::min(want, 8u);
			wpb = std::max<uint32_t>(1, (uint32_t) std::min<size_t>(want, ((size_t) maxlds - 8 * sizeof(TeamCtl)) / a.wave_bytes));
			if (wpb < 2) team = false;
		}
	}
	if (!team) { wpb = 4; while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1; }
	a.off_ctl = (uint32_t) ((size_t) wpb * a.wave_bytes);
	const size_t lds = (size_t) wpb * a.wave_bytes + (team ? wpb * sizeof(TeamCtl) : 0);
	const bool lean = narrow5 && !team && !w->pops_next && !w->evals_next && !w->times_next && knob(K_LEAN, 1) != 0;
	search_kernel_t kern = pick_search_kernel(func_code, a.kiters, rreg, team, narrow5, lean);
	if (!kern) return fail(HNSW_GPU_ERR_INTERNAL, "no kernel for this configuration");
	{
		static const char *const shapes[4] = { "Shape2x4", "Shape4x2", "Shape8x2", "Shape12x2" };
		const char *shp = shapes[shape_index(a.kiters)];
#ifdef HNSW_EXPERIMENT
		if (shape_index(a.kiters) == 3 && knob(K_SHAPE_12X1, 0)) shp = "Shape12x1";
#endif
		if (narrow5 && !team) shp = "Shape2x2";
		if (rreg < 0) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_beam<%d, pgemb::%s, %d, %s, %s>", func_code, shp, -rreg, team ? "true" : "false", lean ? "true" : "false");
		else if (rreg == 3) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_wide<%d, pgemb::%s>", (int) ix->meta.dist_func, shp);
		else if (rreg >= 2) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_reg<%d, pgemb::%s, %d>", (int) ix->meta.dist_func, shp, rreg);
		else snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_lds<%d, pgemb::%s, %s>", (int) ix->meta.dist_func, shp, rreg == 1 ? "true" : "false");
	}
	if (lds > 48 * 1024)
		HIPCHK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	int per_cu = 0;
	HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int) (wpb * 64), lds));
	if (per_cu < 1) per_cu = 1;
	if (knob(K_BLOCKS_PER_CU, 0) > 0) per_cu = std::min(per_cu, (int) knob(K_BLOCKS_PER_CU, 0));
	size_t blocks = std::min<size_t>((nq + wpb - 1) / wpb, (size_t) per_cu * ix->num_cu);
	a.team_mains = wpb;
	if (team && nq < (size_t) per_cu * ix->num_cu * wpb)
	{
		// fewer queries than resident waves: spread them over the blocks, the other waves of a block start as helpers
		blocks = std::min<size_t>(nq, (size_t) per_cu * ix->num_cu);
		a.team_mains = (uint32_t) std::min<size_t>(wpb, (nq + blocks - 1) / blocks);
		// ... unless the caller knows better: a host that keeps SEVERAL small launches in flight (the batching server's lanes) says
		// how many waves of a block should walk — one walk per 8-wave block is the latency shape of a lone launch; six such launches
		// of 190 queries want 9 000 waves of a device that holds 2 048, i.e. at most 256 walks run at a time however many wait
		// (profiles/r4d_server_sweep.txt: the server's 0.54 M q/s ceiling is exactly 256 walks of 0.47 ms)
		if (w->walkers_hint > a.team_mains)
		{
			a.team_mains = std::min<uint32_t>(wpb, w->walkers_hint);
			blocks = std::min<size_t>((nq + a.team_mains - 1) / a.team_mains, (size_t) per_cu * ix->num_cu);
		}
	}
	// (experiment knob: walking waves per block of a team launch — the others help from the start; scripts/exp_spec_ab.py)
#ifdef HNSW_EXPERIMENT
	if (team && knob(K_TEAM_MAINS, 0) > 0) a.team_mains = std::min<uint32_t>(a.team_mains, (uint32_t) knob(K_TEAM_MAINS, 0));
#endif
	if (stream_launch)
	{
		// a resident launch fed by the host (device_search.h, "Stream mode"): exactly the blocks the device holds at once — block 0 is the
		// doorbell and must be resident for any other block to make progress
		if (!team) { w->stream_host_next = nullptr; return fail(HNSW_GPU_ERR_ARG, "a stream needs the team form of the beam kernel (ef <= 256, or <= 512 on wide rows)"); }
		blocks = std::max<size_t>(2, (size_t) per_cu * ix->num_cu);
		a.team_mains = std::min<uint32_t>(wpb, std::max<uint32_t>(1u, w->stream_walkers_next));
		a.stream_host = w->stream_host_next; a.stream_dev = w->stream_dev_next; a.stream_ring = w->stream_ring_next;
		a.stream_light = knob(K_STREAM_LIGHT, 1) != 0 ? 1u : 0u;
		w->stream_host_next = nullptr; w->stream_dev_next = nullptr;
	}
	// (test knob: fewer blocks than the launch would get, so that the waves with queries take SEVERAL each through the
	// ticket counter while their siblings help — the schedule of a small launch whose other blocks start late,
	// tests/experiments/team_second_walk_stress.py)
	if (knob(K_MAX_BLOCKS, 0) > 0) blocks = std::min<size_t>(blocks, (size_t) knob(K_MAX_BLOCKS, 0));

	// workspace: one bitmap + log per resident wave
	const size_t words = std::max<size_t>(1, (ix->cap + 31) / 32);   // by capacity: stable while the index grows
	size_t max_slots = std::max<size_t>(wpb, VIS_BUDGET_BYTES / (words * 4));
	if (rreg == 1 || rreg == 3) max_slots = std::max<size_t>(wpb, std::min(max_slots, SET_BUDGET_BYTES / (a.set_stride * 8)));
	if (blocks * wpb > max_slots) blocks = std::max<size_t>(1, max_slots / wpb);
	const size_t slots = blocks * wpb;
	const uint32_t logcap = 8192;
	if (slots > w->vis_slots || words != w->vis_words)
	{
		if (w->vis) (void) hipFree(w->vis);
		if (w->vlog) (void) hipFree(w->vlog);
		w->vis = nullptr; w->vlog = nullptr; w->vis_slots = 0;
		HIPCHK(hipMalloc(&w->vis, slots * words * 4));
		HIPCHK(hipMalloc(&w->vlog, slots * (size_t) logcap * 4));
		HIPCHK(hipMemsetAsync(w->vis, 0, slots * words * 4, stream));
		w->vis_slots = slots; w->vis_words = words; w->logcap = logcap;
	}
	if (__atomic_load_n(&w->abort_sent, __ATOMIC_SEQ_CST))
	{
		// the previous launch of this workspace was asked to end early: its waves left their bitmaps as they were
		fprintf(stderr, "pg_embedding_amd: the previous search launch of this workspace (%s) was asked to end early (abort word): the queries it did "
				"not answer have count HNSW_GPU_COUNT_ABORTED; the workspace is re-zeroed\n", w->kname);
		HIPCHK(hipStreamSynchronize(stream));
		if (stream) HIPCHK(hipStreamSynchronize(nullptr));
		if (w->vis) HIPCHK(hipMemset(w->vis, 0, w->vis_slots * w->vis_words * 4));
		__atomic_store_n(w->abort_host, 0u, __ATOMIC_SEQ_CST);
		__atomic_store_n(&w->abort_sent, 0, __ATOMIC_SEQ_CST);
	}
	a.health = w->health; a.abort_word = w->abort_host;
	a.vis = w->vis; a.vis_words = words; a.vlog = w->vlog; a.logcap = w->logcap;
	if (ucap && slots * ucap > w->beam_keys)
	{
		if (w->beam) (void) hipFree(w->beam);
		w->beam = nullptr; w->beam_keys = 0;
		HIPCHK(hipMalloc(&w->beam, slots * ucap * 8));
		w->beam_keys = slots * ucap;
	}
	a.beam_scratch = w->beam;
	if (rreg == 1 || rreg == 3)
	{
		const size_t keys = slots * a.set_stride;
		if (keys > w->set_keys)
		{
			if (w->sets) (void) hipFree(w->sets);
			w->sets = nullptr; w->set_keys = 0;
			HIPCHK(hipMalloc(&w->sets, keys * 8));
			w->set_keys = keys;
		}
		a.set_scratch = w->sets;
	}
	a.ticket = w->ticket;
#ifdef HNSW_EXPERIMENT
	if (knob(K_TEAM_COUNTERS, 0))                           // (diagnostic builds only: build.py variant ... HNSW_HOP_STAMPS / HNSW_TEAM_COUNTERS)
	{
		if (!w->team_dbg) HIPCHK(hipMalloc(&w->team_dbg, 64));
		HIPCHK(hipMemsetAsync(w->team_dbg, 0, 64, stream));
		a.team_dbg = w->team_dbg;
	}
#endif
	a.done = w->done_next;
	w->done_next = nullptr;
	a.out_pops = w->pops_next; a.pops_cap = w->pops_cap_next;
	w->pops_next = nullptr; w->pops_cap_next = 0;
	a.out_evals = w->evals_next; a.evals_cap = w->evals_cap_next; a.out_times = w->times_next;
	w->evals_next = nullptr; w->evals_cap_next = 0; w->times_next = nullptr;
	HIPCHK(hipMemsetAsync(w->ticket, 0, 8, stream));

	const int evi = (int) (w->launches % SearchWs::EV_RING);
	HIPCHK(hipEventRecord(w->ev0[evi], stream));
	// (a stream is resident by design: the library's watchdog does not time it — its host stops it, hnsw_gpu_stream_close)
	__atomic_store_n(&w->busy_since_ms, stream_launch ? (int64_t) 0 : now_ms(), __ATOMIC_SEQ_CST);
	hipLaunchKernelGGL(kern, dim3((uint32_t) blocks), dim3(wpb * 64), lds, stream, a);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(w->ev1[evi], stream));
	__atomic_store_n(&w->launches, w->launches + 1, __ATOMIC_SEQ_CST);
	w->last_slots = (uint32_t) slots;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_batch_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										 label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										 void *stream)
{
	return launch_search(ix, ix ? &ix->ws : nullptr, d_queries, ix ? ix->meta.dim : 0, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats, (hipStream_t) stream);
}

// The same launch as hnsw_gpu_search_batch_dev that also writes its evaluation trace: d_evals[i * evals_cap + j] = the j-th row
// query i scored (j < d_stats[2 * i], truncated at evals_cap), d_times[2 * i], [2 * i + 1] = the device's constant-rate clock
// (100 MHz) at the start of query i and at the end of its walk.  Measurement only (bench.py: replay roof, reuse distances).
extern "C" int hnsw_gpu_search_traced_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										  label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										  idx_t *d_evals, size_t evals_cap, uint64_t *d_times, void *stream)
{
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (!d_evals || evals_cap == 0 || evals_cap > 0xFFFFFFFFull || !d_stats) return fail(HNSW_GPU_ERR_ARG, "trace buffers missing");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	ix->ws.evals_next = d_evals; ix->ws.evals_cap_next = (uint32_t) evals_cap; ix->ws.times_next = d_times;
	const int rc = launch_search(ix, &ix->ws, d_queries, ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats, (hipStream_t) stream);
	ix->ws.evals_next = nullptr; ix->ws.evals_cap_next = 0; ix->ws.times_next = nullptr;
	return rc;
}

extern "C" int hnsw_gpu_search_base_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										idx_t *d_idx, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										void *stream)
{
	return launch_search(ix, ix ? &ix->ws : nullptr, d_queries, ix ? ix->meta.dim : 0, nq, ef, 1, nullptr, d_idx, d_dists, d_counts, d_stats, (hipStream_t) stream);
}

// Poll a completion flag the kernel stores into pinned host memory.  0 = set; otherwise an error: the kernel ended
// without storing it, or it has not stored it within two minutes (HNSW_GPU_POLL_LIMIT_S; a walk is under a
// millisecond: the device is hung, and polling for ever would hang the caller with it).
static int poll_limit_s()
{
	knobs_init();
	return knob(K_POLL_LIMIT_S, 0) > 0 ? (int) knob(K_POLL_LIMIT_S, 0) : 120;
}

// `w` = the search workspace whose launch is waited for, or nullptr when the wait is for kernels that do not read an abort word
// (the insert kernels): on a time-out only THAT workspace is asked to end — other mirrors, contexts and shards of the process keep
// their launches (an abort makes a launch's outputs undefined).
static int poll_done_flag(const volatile uint32_t *flag, const char *what, SearchWs *w)
{
	uint64_t spins = 0;
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while (*flag == 0)
	{
		__builtin_ia32_pause();
		if ((++spins & 0xFFFF) == 0)
		{
			if (hipStreamQuery(nullptr) != hipErrorNotReady)
			{
				HIPCHK(hipStreamSynchronize(nullptr));                  // the kernel is gone: either it has just stored the flag, or it died
				if (*flag == 0) return fail(HNSW_GPU_ERR_INTERNAL, "search kernel ended without completing %s", what);
				break;
			}
			struct timespec t1;
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (t1.tv_sec - t0.tv_sec > poll_limit_s())
			{
				// ask THIS launch to end (every wave looks at the abort word between queries and every 256 hops), so that the
				// device is usable again even though this call fails
				if (w) { std::lock_guard<std::mutex> g(g_ws_mu); (void) abort_ws_locked(w); }
				return fail(HNSW_GPU_ERR_INTERNAL, "kernel did not complete %s within %d s%s", what, poll_limit_s(), w ? " (its search launch was asked to end)" : "");
			}
		}
	}
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_batch(hnsw_gpu_index *ix, const coord_t *queries, size_t nq, size_t ef,
									 label_t *labels, dist_t *dists, uint32_t *counts)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	HIPCHK(hipSetDevice(ix->device));
	const size_t dim = ix->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	int rc;
	// A few queries per call — the reference's own shape is ONE (hnsw_search, embedding.c:317) — are pure latency: the
	// walk is ~0.45 ms and four blocking copies plus a stream wait added ~50 us to it.  Here the kernel reads the
	// queries from pinned host memory, writes results and per-query completion flags (system-scope release, the
	// server's streamed-completion mechanism) straight back into it, and the calling core polls the flags: no copy
	// engine, no interrupt wake-up.  The launch stays on the default stream, so whatever touches this mirror next is
	// ordered behind the kernel's last instruction, not behind the flags.
	const size_t fb = round_up(nq * 4, 256);
	if (nq <= 16 && qb + lb + db + cb + fb <= ((size_t) 4 << 20) && (knobs_init(), knob(K_NO_POLL, 0) == 0))
	{
		if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes these buffers
		if (ix->pin_bytes < qb + lb + db + cb + fb)
		{
			if (ix->pin) (void) hipHostFree(ix->pin);
			ix->pin = nullptr; ix->pin_bytes = 0;
			HIPCHK(hipHostMalloc((void **) &ix->pin, qb + lb + db + cb + fb, hipHostMallocDefault));
			ix->pin_bytes = qb + lb + db + cb + fb;
		}
		char *h = ix->pin;
		float *hq = (float *) h; uint64_t *hl = (uint64_t *) (h + qb); float *hd = (float *) (h + qb + lb);
		uint32_t *hc = (uint32_t *) (h + qb + lb + db);
		volatile uint32_t *hf = (volatile uint32_t *) (h + qb + lb + db + cb);
		memcpy(hq, queries, nq * dim * 4);
		for (size_t i = 0; i < nq; i++) hf[i] = 0;
		ix->ws.done_next = (uint32_t *) hf;
		rc = launch_search(ix, &ix->ws, hq, dim, nq, ef, 0, hl, nullptr, hd, hc, nullptr, nullptr);
		ix->ws.done_next = nullptr;
		if (rc) return rc;
		for (size_t i = 0; i < nq; i++)
		{
			rc = poll_done_flag(hf + i, "a query", &ix->ws);
			if (rc) return rc;
		}
		memcpy(labels, hl, nq * ef * 8);
		if (dists) memcpy(dists, hd, nq * ef * 4);
		memcpy(counts, hc, nq * 4);
		return HNSW_GPU_OK;
	}
	rc = ensure_scratch(ix, qb + lb + db + cb);
	if (rc) return rc;
	char *p = (char *) ix->scratch;
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	if (!ix->hb0) { HIPCHK(hipEventCreate(&ix->hb0)); HIPCHK(hipEventCreate(&ix->hb1)); }
	ix->hb_valid = false;
	HIPCHK(hipEventRecord(ix->hb0, nullptr));
	HIPCHK(hipMemcpy(dq, queries, nq * dim * 4, hipMemcpyHostToDevice));
	rc = launch_search(ix, &ix->ws, dq, dim, nq, ef, 0, dl, nullptr, dd, dc, nullptr, nullptr);
	if (rc) return rc;
	HIPCHK(hipMemcpy(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost));
	if (dists) HIPCHK(hipMemcpy(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(counts, dc, nq * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipEventRecord(ix->hb1, nullptr));
	ix->hb_valid = true;
	for (size_t i = 0; i < nq; i++)
		if (counts[i] == ABORTED_COUNT)
			return fail(HNSW_GPU_ERR_INTERNAL, "the search launch was asked to end early (abort word): query %zu has no result", i);
	return HNSW_GPU_OK;
}

// One query with its walk: results as hnsw_gpu_search_batch gives them, plus the sequence of elements the walk expanded
// (hnswalg.cpp:73) and its evaluation count.  Host pointers; the polled zero-copy mechanics of the few-queries path.
// Three steps so that a caller can consume the sequence WHILE the walk runs (the kernel stores each pop with system
// scope into pinned host memory): begin = launch, poll = the pops that have become visible since the last poll,
// end = wait + results.  One trace at a time per mirror, from one thread; no library lock is held between the steps
// (the caller may run host callbacks that leave by longjmp in between: a trace that is never ended is waited for by the
// next begin).
static const uint32_t POP_NONE = 0xFFFFFFFFu;

struct TraceLayout { size_t qb, lb, db, cb, sb, pb, fb; };
static TraceLayout trace_layout(size_t dim, size_t ef, size_t pops_cap)
{
	TraceLayout t;
	t.qb = round_up(dim * 4, 256); t.lb = round_up(ef * 8, 256); t.db = round_up(ef * 4, 256); t.cb = 256; t.sb = 256;
	t.pb = round_up(pops_cap * 4, 256); t.fb = 256;
	return t;
}

extern "C" int hnsw_gpu_search_trace_begin(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, size_t pops_cap)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (!query || pops_cap == 0) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	if (pops_cap > ((size_t) 1 << 24)) return fail(HNSW_GPU_ERR_ARG, "pops_cap %zu too large", pops_cap);
	HIPCHK(hipSetDevice(ix->device));
	if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes its buffers
	const size_t dim = ix->meta.dim;
	const TraceLayout t = trace_layout(dim, ef, pops_cap);
	const size_t need = t.qb + t.lb + t.db + t.cb + t.sb + t.pb + t.fb;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, need, hipHostMallocDefault));
		ix->pin_bytes = need;
	}
	char *h = ix->pin;
	float *hq = (float *) h; uint64_t *hl = (uint64_t *) (h + t.qb); float *hd = (float *) (h + t.qb + t.lb);
	uint32_t *hc = (uint32_t *) (h + t.qb + t.lb + t.db), *hs = (uint32_t *) (h + t.qb + t.lb + t.db + t.cb),
			 *hp = (uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb);
	volatile uint32_t *hf = (volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	memcpy(hq, query, dim * 4);
	memset(hp, 0xFF, pops_cap * 4);                 // POP_NONE: a slot the walk has not reached yet
	hf[0] = 0;
	ix->ws.done_next = (uint32_t *) hf;
	ix->ws.pops_next = hp; ix->ws.pops_cap_next = (uint32_t) pops_cap;
	int rc = base ? launch_search(ix, &ix->ws, hq, dim, 1, ef, 1, nullptr, (uint32_t *) hl, hd, hc, hs, nullptr)
				  : launch_search(ix, &ix->ws, hq, dim, 1, ef, 0, hl, nullptr, hd, hc, hs, nullptr);
	ix->ws.done_next = nullptr; ix->ws.pops_next = nullptr; ix->ws.pops_cap_next = 0;
	if (rc) return rc;
	ix->trace_active = true; ix->trace_ef = ef; ix->trace_base = base; ix->trace_cap = pops_cap; ix->trace_seen = 0;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace_poll(hnsw_gpu_index *ix, idx_t *pops, size_t max, size_t *got, int *finished)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !pops || !got || !finished) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (!ix->trace_active) return fail(HNSW_GPU_ERR_ARG, "no trace in flight");
	const TraceLayout t = trace_layout(ix->meta.dim, ix->trace_ef, ix->trace_cap);
	char *h = ix->pin;
	const volatile uint32_t *hp = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb);
	const volatile uint32_t *hf = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	const bool done = hf[0] != 0;                   // read BEFORE the scan: everything the walk stored precedes the flag
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	size_t k = 0;
	while (k < max && ix->trace_seen < ix->trace_cap)
	{
		const uint32_t v = hp[ix->trace_seen];
		if (v == POP_NONE) break;
		pops[k++] = v;
		ix->trace_seen++;
	}
	*got = k;
	*finished = (done && (ix->trace_seen >= ix->trace_cap || hp[ix->trace_seen] == POP_NONE)) ? 1 : 0;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace_end(hnsw_gpu_index *ix, label_t *labels, dist_t *dists, uint32_t *count, uint32_t *npops,
										 uint32_t *nevals)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !labels || !count || !npops) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (!ix->trace_active) return fail(HNSW_GPU_ERR_ARG, "no trace in flight");
	HIPCHK(hipSetDevice(ix->device));
	const size_t ef = ix->trace_ef;
	const TraceLayout t = trace_layout(ix->meta.dim, ef, ix->trace_cap);
	char *h = ix->pin;
	const uint64_t *hl = (const uint64_t *) (h + t.qb); const float *hd = (const float *) (h + t.qb + t.lb);
	const uint32_t *hc = (const uint32_t *) (h + t.qb + t.lb + t.db), *hs = (const uint32_t *) (h + t.qb + t.lb + t.db + t.cb);
	const volatile uint32_t *hf = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	{
		const int prc = poll_done_flag(hf, "the traced query", &ix->ws);
		ix->trace_active = false;
		if (prc) return prc;
	}
	if (ix->trace_base) { const uint32_t *hi = (const uint32_t *) hl; for (size_t i = 0; i < ef; i++) labels[i] = hi[i]; }
	else memcpy(labels, hl, ef * 8);
	if (dists) memcpy(dists, hd, ef * 4);
	*count = hc[0];
	*npops = hs[1];
	if (nevals) *nevals = hs[0];
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, label_t *labels, dist_t *dists,
									 uint32_t *count, idx_t *pops, size_t pops_cap, uint32_t *npops, uint32_t *nevals)
{
	if (!ix || !pops || !npops) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::unique_lock<std::recursive_mutex> lock_(ix->mu);          // the three steps as one
	int rc = hnsw_gpu_search_trace_begin(ix, query, ef, base, pops_cap);
	if (rc) return rc;
	rc = hnsw_gpu_search_trace_end(ix, labels, dists, count, npops, nevals);
	if (rc) return rc;
	const TraceLayout t = trace_layout(ix->meta.dim, ef, pops_cap);
	memcpy(pops, ix->pin + t.qb + t.lb + t.db + t.cb + t.sb, std::min<size_t>(*npops, pops_cap) * 4);
	return HNSW_GPU_OK;
}

static int ws_search_ms(int device, SearchWs *w, unsigned back, float *ms)
{
	if (back >= (unsigned) SearchWs::EV_RING || (uint64_t) back >= w->launches)
		return fail(HNSW_GPU_ERR_ARG, "no record of the search launch %u launches ago", back);
	HIPCHK(hipSetDevice(device));
	const int evi = (int) ((w->launches - 1 - back) % SearchWs::EV_RING);
	HIPCHK(hipEventSynchronize(w->ev1[evi]));
	HIPCHK(hipEventElapsedTime(ms, w->ev0[evi], w->ev1[evi]));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_ms(hnsw_gpu_index *ix, unsigned back, float *ms)
{
	if (!ix || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return ws_search_ms(ix->device, &ix->ws, back, ms);
}

extern "C" int hnsw_gpu_last_search_ms(hnsw_gpu_index *ix, float *ms) { return hnsw_gpu_search_ms(ix, 0, ms); }

// Where the time of the last hnsw_gpu_search_batch call (host pointers, copy path: more than 16 queries) went on the device:
// out[0] = upload of the queries, out[1] = the search kernel, out[2] = download of labels / distances / counts (milliseconds,
// HIP events on the default stream around the three steps).  SURVEY.md §8(d): "report H2D separately".
extern "C" int hnsw_gpu_last_batch_ms(hnsw_gpu_index *ix, float out[3])
{
	if (!ix || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	if (!ix->hb_valid || ix->ws.launches == 0) return fail(HNSW_GPU_ERR_ARG, "no host-pointer batch call (copy path) has completed on this mirror");
	HIPCHK(hipSetDevice(ix->device));
	const int evi = (int) ((ix->ws.launches - 1) % SearchWs::EV_RING);
	HIPCHK(hipEventSynchronize(ix->hb1));
	HIPCHK(hipEventElapsedTime(&out[0], ix->hb0, ix->ws.ev0[evi]));
	HIPCHK(hipEventElapsedTime(&out[1], ix->ws.ev0[evi], ix->ws.ev1[evi]));
	HIPCHK(hipEventElapsedTime(&out[2], ix->ws.ev1[evi], ix->hb1));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_search_kernel(hnsw_gpu_index *ix, char *buf, size_t len)
{
	if (!ix || !buf || len == 0) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	snprintf(buf, len, "%s", ix->ws.kname);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_team_counters(hnsw_gpu_index *ix, uint32_t *out8)
{
	if (!ix || !out8) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	memset(out8, 0, 64);
	if (!ix->ws.team_dbg) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(out8, ix->ws.team_dbg, 64, hipMemcpyDeviceToHost));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_abort(hnsw_gpu_index *ix)
{
	// no ix->mu here: the thread that holds it may be the one waiting for the launch this call is meant to end
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	std::lock_guard<std::mutex> g(g_ws_mu);
	return abort_ws_locked(&ix->ws) ? HNSW_GPU_OK : fail(HNSW_GPU_ERR_INTERNAL, "the workspace has no abort word");
}

extern "C" int hnsw_gpu_index_health(hnsw_gpu_index *ix, uint32_t *out8)
{
	if (!ix || !out8) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipMemcpy(out8, ix->ws.health, 32, hipMemcpyDeviceToHost));
	out8[0] = __atomic_load_n(ix->ws.abort_host, __ATOMIC_SEQ_CST);
	out8[5] = __atomic_load_n(&ix->ws.abort_requests, __ATOMIC_SEQ_CST);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_search_slots(hnsw_gpu_index *ix, uint32_t *slots)
{
	if (!ix || !slots) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	*slots = ix->ws.last_slots;
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// batched distances (hnsw_dist_func over many rows)
// ------------------------------------------------------------------------------------
template <int FUNC>
__global__ __launch_bounds__(256) void dist_batch_kernel(const float *__restrict__ q, const float *__restrict__ rows,
														 uint32_t nrows, uint32_t dim, uint32_t stride, uint32_t nchunks,
														 uint32_t kiters, uint32_t qpad_floats, float *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float *qf = reinterpret_cast<float *>(smem);
	const float4 *q4 = reinterpret_cast<const float4 *>(smem);
	float *sums = reinterpret_cast<float *>(smem + (size_t) qpad_floats * 4) + (threadIdx.x >> 6) * 128;   // per wave
	for (uint32_t e = threadIdx.x; e < qpad_floats; e += blockDim.x)
	{
		const float t = q[e < dim ? e : dim - 1];
		qf[e] = (e < dim) ? t : 0.f;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, nchunks, kiters, lane);
	for (uint32_t base = wave * 64; base < nrows; base += nwaves * 64)
	{
		const uint32_t cnt = min(64u, nrows - base);
		auto direct = [base](uint32_t r) { return base + r; };
		score_rows<FUNC, 4, 2>(rows, stride, q4, nchunks, kiters, direct, cnt, sums, lane);
		wave_sync();
		const float d = finish_dist<FUNC>(sums[lane], sums[OUT2 + lane], qnorm);
		if ((uint32_t) lane < cnt) out[base + lane] = d;
		wave_sync();
	}
}

extern "C" int hnsw_gpu_dist_batch_dev(dist_func_t func, const coord_t *d_q, const coord_t *d_rows, size_t nrows,
									   size_t dim, size_t row_stride, dist_t *d_out, void *stream)
{
	if (nrows == 0) return HNSW_GPU_OK;
	if (!d_q || !d_rows || !d_out) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if ((int) func < 0 || (int) func > 2) return fail(HNSW_GPU_ERR_ARG, "bad dist_func %d", (int) func);
	if (dim == 0 || row_stride < dim || (row_stride & 3) || (((uintptr_t) d_rows) & 15))
		return fail(HNSW_GPU_ERR_ARG, "rows must be 16-byte aligned with stride %% 4 == 0 and stride >= dim");
	if (nrows >= 0xFFFFFFF0ull) return fail(HNSW_GPU_ERR_ARG, "too many rows");
	const uint32_t nchunks = (uint32_t) (row_stride / 4), kiters = (nchunks + 15) / 16;
	const uint32_t qpad = (uint32_t) round_up(kiters, 4) * 64;
	const size_t lds = (size_t) qpad * 4 + 4 * 128 * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "dim %zu too large", dim);
	const uint32_t blocks = (uint32_t) std::min<size_t>((nrows + 255) / 256, 256 * 8);
	hipStream_t s = (hipStream_t) stream;
	switch ((int) func)
	{
		case F_L2:
			hipLaunchKernelGGL(dist_batch_kernel<F_L2>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
		case F_COSINE:
			hipLaunchKernelGGL(dist_batch_kernel<F_COSINE>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
		default:
			hipLaunchKernelGGL(dist_batch_kernel<F_MANHATTAN>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
	}
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_dist_batch(dist_func_t func, const coord_t *q, const coord_t *rows, size_t nrows, size_t dim,
								   dist_t *out)
{
	if (nrows == 0) return HNSW_GPU_OK;
	if (!q || !rows || !out) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (dim == 0) return fail(HNSW_GPU_ERR_ARG, "dim is 0");
	if (hnsw_gpu_device_count() <= 0) return fail(HNSW_GPU_ERR_NODEVICE, "no HIP device visible (this library has no CPU path)");
	const size_t stride = round_up(dim, 4);
	// Small calls — the SQL operators hand over ONE pair per call (embedding.c:1037) — go through a
	// per-thread pinned staging area that the kernel reads and writes directly: no allocation, no copy
	// engine, one launch + one stream wait.
	const size_t small_bytes = (1 + nrows) * stride * 4 + round_up(nrows * 4, 16);
	if (small_bytes <= ((size_t) 256 << 10))
	{
		static thread_local char *pin = nullptr;
		static thread_local size_t pin_bytes = 0;
		static thread_local hipStream_t pin_stream = nullptr;
		static thread_local int pin_device = -1;
		int dev = 0;
		HIPCHK(hipGetDevice(&dev));
		if (pin_device != dev || pin_bytes < small_bytes)
		{
			if (pin) (void) hipHostFree(pin);
			if (pin_stream) (void) hipStreamDestroy(pin_stream);
			pin = nullptr; pin_bytes = 0; pin_stream = nullptr; pin_device = -1;
			HIPCHK(hipHostMalloc((void **) &pin, (size_t) 256 << 10, hipHostMallocDefault));
			HIPCHK(hipStreamCreateWithFlags(&pin_stream, hipStreamNonBlocking));
			pin_bytes = (size_t) 256 << 10;
			pin_device = dev;
		}
		float *hq = (float *) pin, *hr = hq + stride, *ho = (float *) (pin + (1 + nrows) * stride * 4);
		memcpy(hq, q, dim * 4);
		for (size_t d = dim; d < stride; d++) hq[d] = 0.f;
		for (size_t r = 0; r < nrows; r++)
		{
			memcpy(hr + r * stride, rows + r * dim, dim * 4);
			for (size_t d = dim; d < stride; d++) hr[r * stride + d] = 0.f;
		}
		int rc2 = hnsw_gpu_dist_batch_dev(func, hq, hr, nrows, dim, stride, ho, pin_stream);
		if (rc2) return rc2;
		HIPCHK(hipStreamSynchronize(pin_stream));
		memcpy(out, ho, nrows * 4);
		return HNSW_GPU_OK;
	}
	float *dq = nullptr, *dr = nullptr, *dout = nullptr;
	hipError_t e = hipSuccess;
	int rc = HNSW_GPU_OK;
	if ((e = hipMalloc(&dq, dim * 4)) != hipSuccess || (e = hipMalloc(&dr, nrows * stride * 4)) != hipSuccess ||
		(e = hipMalloc(&dout, nrows * 4)) != hipSuccess)
		rc = fail(HNSW_GPU_ERR_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
	if (!rc && stride != dim && (e = hipMemset(dr, 0, nrows * stride * 4)) != hipSuccess) rc = fail(HNSW_GPU_ERR_HIP, "memset failed");
	if (!rc && ((e = hipMemcpy(dq, q, dim * 4, hipMemcpyHostToDevice)) != hipSuccess ||
				(e = hipMemcpy2D(dr, stride * 4, rows, dim * 4, dim * 4, nrows, hipMemcpyHostToDevice)) != hipSuccess))
		rc = fail(HNSW_GPU_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
	if (!rc) rc = hnsw_gpu_dist_batch_dev(func, dq, dr, nrows, dim, stride, dout, nullptr);
	if (!rc && (e = hipMemcpy(out, dout, nrows * 4, hipMemcpyDeviceToHost)) != hipSuccess)
		rc = fail(HNSW_GPU_ERR_HIP, "download failed: %s", hipGetErrorString(e));
	if (dq) (void) hipFree(dq);
	if (dr) (void) hipFree(dr);
	if (dout) (void) hipFree(dout);
	return rc;
}

// ------------------------------------------------------------------------------------
// exhaustive k-NN with the same distance code (recall ground truth)
// ------------------------------------------------------------------------------------
// grid = (splits, nq); each wave scans a contiguous slice of the rows for one query and keeps a
// sorted top-k of (ord(dist)<<32 | idx) keys in LDS; partial lists are merged by topk_merge_kernel.
template <int FUNC>
__global__ __launch_bounds__(256) void bruteforce_kernel(const float *__restrict__ vec, uint32_t n, uint32_t dim,
														 uint32_t stride, uint32_t nchunks, uint32_t kiters,
														 uint32_t qpad_floats, const float *__restrict__ queries,
														 uint32_t k, uint64_t *__restrict__ part /* [nq][splits*4][k] */)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const uint32_t qi = blockIdx.y;
	float *qf = reinterpret_cast<float *>(smem);
	const float4 *q4 = reinterpret_cast<const float4 *>(smem);
	for (uint32_t e = threadIdx.x; e < qpad_floats; e += blockDim.x) qf[e] = (e < dim) ? queries[(size_t) qi * dim + e] : 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	uint64_t *top = reinterpret_cast<uint64_t *>(smem + (size_t) qpad_floats * 4) + (size_t) wib * (k + 1);
	float *sums = reinterpret_cast<float *>(smem + (size_t) qpad_floats * 4 + (size_t) 4 * (k + 1) * 8) + wib * 128;
	const uint32_t nw = gridDim.x * 4, w = blockIdx.x * 4 + wib;
	const uint32_t lo = (uint32_t) ((uint64_t) n * w / nw), hi = (uint32_t) ((uint64_t) n * (w + 1) / nw);
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, nchunks, kiters, lane);
	uint32_t tsize = 0;
	uint64_t worst = ~0ull;
	for (uint32_t base = lo; base < hi; base += 64)
	{
		const uint32_t cnt = min(64u, hi - base);
		auto direct = [base](uint32_t r) { return base + r; };
		score_rows<FUNC, 4, 2>(vec, stride, q4, nchunks, kiters, direct, cnt, sums, lane);
		wave_sync();
		const float dl = finish_dist<FUNC>(sums[lane], sums[OUT2 + lane], qnorm);
		const uint64_t kl = ((uint64_t) ord_f32(dl) << 32) | (base + lane);
		// only rows that can enter the current top-k are visited one by one
		uint64_t todo = __ballot((uint32_t) lane < cnt && (tsize < k || kl < worst));
		while (todo)
		{
			const uint32_t r = (uint32_t) __builtin_ctzll(todo);
			todo &= todo - 1;
			const uint64_t key = readlane_u64(kl, r);
			if (tsize < k || key < worst)
			{
				tsize = sorted_insert(top, tsize, key, k, lane);
				worst = top[tsize - 1];
			}
		}
		wave_sync();
	}
	uint64_t *dst = part + ((size_t) qi * nw + w) * k;
	for (uint32_t i = lane; i < k; i += 64) dst[i] = (i < tsize) ? top[i] : ~0ull;
}

// One wave per query: merge `nlists` ascending key lists of length k into the k smallest.
__global__ __launch_bounds__(64) void key_merge_kernel(const uint64_t *__restrict__ part, uint32_t nlists, uint32_t k,
													   uint32_t *__restrict__ out_idx, float *__restrict__ out_dist)
{
	const uint32_t qi = blockIdx.x;
	const int lane = threadIdx.x;
	const uint64_t *src = part + (size_t) qi * nlists * k;
	const uint32_t total = nlists * k;
	for (uint32_t x = lane; x < total; x += 64)
	{
		const uint32_t l = x / k;
		const uint64_t key = src[x];
		if (key == ~0ull) continue;
		uint32_t rank = x - l * k;
		for (uint32_t m = 0; m < nlists && rank < k; m++)
		{
			if (m == l) continue;
			const uint64_t *o = src + (size_t) m * k;
			uint32_t lo = 0, hi = k;                       // number of keys in list m below `key` (keys are unique)
			while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (o[mid] < key) lo = mid + 1; else hi = mid; }
			rank += lo;
		}
		if (rank < k)
		{
			out_idx[(size_t) qi * k + rank] = (uint32_t) key;
			if (out_dist) out_dist[(size_t) qi * k + rank] = unord_f32((uint32_t) (key >> 32));
		}
	}
}

__global__ void fill_u32_kernel(uint32_t *p, size_t n, uint32_t v)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

static int bruteforce_prefix(hnsw_gpu_index *ix, size_t nrows, const coord_t *d_queries, size_t nq, size_t k, idx_t *d_idx,
							 dist_t *d_dists, void *stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_queries || !d_idx) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (nq == 0) return HNSW_GPU_OK;
	if (k == 0 || k > 1024) return fail(HNSW_GPU_ERR_ARG, "k %zu out of range [1, 1024]", k);
	if (nq > 65535) return fail(HNSW_GPU_ERR_ARG, "at most 65535 queries per call");
	HIPCHK(hipSetDevice(ix->device));
	hipStream_t s = (hipStream_t) stream;
	const uint32_t nchunks = ix->stride / 4, kiters = (nchunks + 15) / 16;
	const uint32_t qpad = (uint32_t) round_up(kiters, 4) * 64;
	uint32_t splits = (uint32_t) std::max<size_t>(1, std::min<size_t>(64, (size_t) (4 * ix->num_cu) / nq));
	splits = (uint32_t) std::min<size_t>(splits, std::max<size_t>(1, nrows / 64));
	const uint32_t nlists = splits * 4;
	const size_t lds = (size_t) qpad * 4 + (size_t) 4 * (k + 1) * 8 + 4 * 128 * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "k/dim too large for brute force");
	int rc = ensure_scratch(ix, nq * nlists * k * 8);
	if (rc) return rc;
	uint64_t *part = (uint64_t *) ix->scratch;
	const size_t tot = nq * k;
	hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t) ((tot + 255) / 256)), dim3(256), 0, s, d_idx, tot, LINK_NONE);
	if (d_dists)
		hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t) ((tot + 255) / 256)), dim3(256), 0, s, (uint32_t *) d_dists, tot,
						   0x7F800000u);
	dim3 grid(splits, (uint32_t) nq);
#define BF_LAUNCH(F)                                                                                                   \
	hipLaunchKernelGGL(bruteforce_kernel<F>, grid, dim3(256), lds, s, ix->vec, (uint32_t) nrows, (uint32_t) ix->meta.dim, \
					   ix->stride, nchunks, kiters, qpad, d_queries, (uint32_t) k, part)
	switch ((int) ix->meta.dist_func)
	{
		case F_L2: BF_LAUNCH(F_L2); break;
		case F_COSINE: BF_LAUNCH(F_COSINE); break;
		default: BF_LAUNCH(F_MANHATTAN); break;
	}
#undef BF_LAUNCH
	hipLaunchKernelGGL(key_merge_kernel, dim3((uint32_t) nq), dim3(64), 0, s, part, nlists, (uint32_t) k, d_idx, d_dists);
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_bruteforce_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k, idx_t *d_idx,
									   dist_t *d_dists, void *stream)
{
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return bruteforce_prefix(ix, ix->n, d_queries, nq, k, d_idx, d_dists, stream);
}

// ------------------------------------------------------------------------------------
// exhaustive k-NN with the dense part on the matrix cores (device_bf_mfma.h)
// ------------------------------------------------------------------------------------
static float g_last_bf_gemm_ms = 0.f;
static unsigned long long g_last_bf_clocks[2] = { 0, 0 };
static int g_last_bf_tile = 0;

// the filter launch for one tile shape (LDS per block: 69 KB for 128 x 128 tiles, 134 KB for 256 x 256; set per call: the attribute is per device)
template <int WM, int NJ>
static int bf_filter_launch(BfArgs &a, uint32_t nq, uint32_t n, hipStream_t s)
{
	using T = BfTile<WM, NJ>;
	a.nqt = (nq + T::TQ - 1) / T::TQ;
	a.nrt = (n + T::TR - 1) / T::TR;
	const uint32_t rgroups = (a.nrt + 7) / 8;
	HIPCHK(hipFuncSetAttribute((const void *) bf_mfma_filter_kernel<WM, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) T::LDS_BYTES));
	hipLaunchKernelGGL((bf_mfma_filter_kernel<WM, NJ>), dim3(rgroups * a.nqt * 8), dim3(T::THREADS), T::LDS_BYTES, s, a);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_bruteforce_mfma_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k,
											idx_t *d_idx, dist_t *d_dists, void *stream_)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_queries || !d_idx) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (nq == 0) return HNSW_GPU_OK;
	if (k == 0 || k > 1024) return fail(HNSW_GPU_ERR_ARG, "k %zu out of range [1, 1024]", k);
	if (nq > 65535) return fail(HNSW_GPU_ERR_ARG, "at most 65535 queries per call");
	const int func = (int) ix->meta.dist_func;
	if (func == F_MANHATTAN || ix->n < 4096)          // not a contraction / too small to matter
		return hnsw_gpu_bruteforce_dev(ix, d_queries, nq, k, d_idx, d_dists, stream_);
	HIPCHK(hipSetDevice(ix->device));
	hipStream_t s = (hipStream_t) stream_;
	const uint32_t n = (uint32_t) ix->n, stride = ix->stride, dim = (uint32_t) ix->meta.dim;
	const uint32_t nchunks = stride / 4, kiters = (nchunks + 15) / 16;

	// |row|^2 cache
	if (ix->xnorm_cap < ix->n)
	{
		if (ix->xnorm) (void) hipFree(ix->xnorm);
		ix->xnorm = nullptr; ix->xnorm_cap = 0; ix->xnorm_n = 0;
		HIPCHK(hipMalloc(&ix->xnorm, ix->cap * sizeof(float)));
		ix->xnorm_cap = ix->cap;
	}
	if (ix->xnorm_n != ix->n)
	{
		hipLaunchKernelGGL(row_norm2_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix->vec, n, stride, ix->xnorm);
		ix->xnorm_n = ix->n;
	}

	const uint32_t cap = 16384;
	const size_t sample = std::min<size_t>(ix->n, std::max<size_t>(8192, (size_t) k * ix->n / 2048));
	// scratch carve
	const size_t o_q = 0;
	const uint32_t qstride = (uint32_t) round_up(stride, BF_TK);        // the filter's query copy: zero padded to whole K steps
	const size_t o_qn = o_q + round_up(nq * qstride * 4, 256);
	const size_t o_sidx = o_qn + round_up(nq * 4, 256);
	const size_t o_sdist = o_sidx + round_up(nq * k * 4, 256);
	const size_t o_bound = o_sdist + round_up(nq * k * 4, 256);
	const size_t o_cnt = o_bound + round_up(nq * 4, 256);
	const size_t o_cand = o_cnt + round_up(nq * 4 + 64, 256);
	const size_t o_clk = o_cand + round_up(nq * (size_t) cap * 4, 256);
	const size_t total = o_clk + 256;
	if (total > ix->bf_bytes)
	{
		if (ix->bf) (void) hipFree(ix->bf);
		ix->bf = nullptr; ix->bf_bytes = 0;
		HIPCHK(hipMalloc(&ix->bf, total));
		ix->bf_bytes = total;
	}
	char *B = (char *) ix->bf;
	float *qpad = (float *) (B + o_q), *qn = (float *) (B + o_qn), *sdist = (float *) (B + o_sdist), *bound = (float *) (B + o_bound);
	uint32_t *sidx = (uint32_t *) (B + o_sidx), *cnt = (uint32_t *) (B + o_cnt), *cand = (uint32_t *) (B + o_cand);
	uint32_t *overflow = cnt + nq;

	// 1. bound per query from a canonical scan of the sample rows
	int rc = bruteforce_prefix(ix, sample, d_queries, nq, k, sidx, sdist, s);
	if (rc) return rc;
	const size_t qtot = nq * (size_t) qstride;
	hipLaunchKernelGGL(pad_queries_kernel, dim3((uint32_t) ((qtot + 255) / 256)), dim3(256), 0, s, d_queries, (uint32_t) nq, dim, qstride, qpad);
	hipLaunchKernelGGL(row_norm2_kernel, dim3((uint32_t) ((nq + 3) / 4)), dim3(256), 0, s, qpad, (uint32_t) nq, qstride, qn);
	// tau_q = sdist[q*k + k-1]: gather with a strided view
	{
		// reuse make_bounds on a compacted tau array: write tau into `bound` first
		hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(1), 0, s, overflow, (size_t) 1, 0u);
		HIPCHK(hipMemcpy2DAsync(bound, 4, sdist + (k - 1), k * 4, 4, nq, hipMemcpyDeviceToDevice, s));
		hipLaunchKernelGGL(make_bounds_kernel, dim3((uint32_t) ((nq + 255) / 256)), dim3(256), 0, s, bound, qn, (uint32_t) nq, func, bound);
	}
	HIPCHK(hipMemsetAsync(cnt, 0, nq * 4, s));

	// 2. the dense contraction + filter
	BfArgs a;
	memset(&a, 0, sizeof(a));
	a.queries = qpad; a.qnorm = qn; a.qbound = bound; a.vec = ix->vec; a.xnorm = ix->xnorm;
	a.nq = (uint32_t) nq; a.n = n; a.stride = stride; a.qstride = qstride; a.ksteps = qstride / BF_TK; a.func = func;
	a.cand = cand; a.cand_cnt = cnt; a.cap = cap; a.clocks = (unsigned long long *) (B + o_clk);
	if (!ix->bf_e0) { HIPCHK(hipEventCreate(&ix->bf_e0)); HIPCHK(hipEventCreate(&ix->bf_e1)); }
	hipEvent_t e0 = ix->bf_e0, e1 = ix->bf_e1;
	HIPCHK(hipEventRecord(e0, s));
	// 256 x 256 tiles when they compute no more padding than 128 x 128 tiles would (an even number of 128-query tiles) and there are
	// tiles enough to fill the device several times over; the same dot products in the same k order either way: the same survivors
	{
		using Big = BfTile<4, 4>;
		const uint64_t nqt_s = (nq + BfTile<BF_WM, BF_NJ>::TQ - 1) / BfTile<BF_WM, BF_NJ>::TQ;
		const uint64_t big_blocks = ((nq + Big::TQ - 1) / Big::TQ) * ((n + Big::TR - 1) / Big::TR);
		// (test knob: 0 = never, < 0 = always, n = at least n blocks; the tests run every case through both tiles)
		const long long min_blocks = knob(K_BF_BIG_MIN_BLOCKS, 2048);
		const bool big = BF_BIG && BF_WM == 2 && BF_NJ == 2 && min_blocks != 0 &&
						 (min_blocks < 0 || (nqt_s % 2 == 0 && big_blocks >= (uint64_t) min_blocks));
		rc = big ? bf_filter_launch<4, 4>(a, (uint32_t) nq, n, s) : bf_filter_launch<BF_WM, BF_NJ>(a, (uint32_t) nq, n, s);
		if (rc) return rc;
		g_last_bf_tile = big ? Big::TQ : BfTile<BF_WM, BF_NJ>::TQ;
	}
	HIPCHK(hipEventRecord(e1, s));

	// 3. canonical re-score of the survivors
	const uint32_t qpadf = (uint32_t) round_up(kiters, 4) * 64;
	const size_t wave_bytes = round_up((size_t) qpadf * 4 + (k + 1) * 8 + 128 * 4, 16);
	const size_t lds = wave_bytes * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "k/dim too large for the rescoring step");
#define RS_LAUNCH(F)                                                                                                      \
	hipLaunchKernelGGL(bf_rescore_kernel<F>, dim3((uint32_t) ((nq + 3) / 4)), dim3(256), lds, s, ix->vec, dim, stride,      \
					   nchunks, kiters, qpadf, d_queries, (uint32_t) nq, cand, cnt, cap, (uint32_t) k, d_idx, d_dists, overflow)
	if (func == F_L2) RS_LAUNCH(F_L2); else RS_LAUNCH(F_COSINE);
#undef RS_LAUNCH
	HIPCHK(hipGetLastError());
	uint32_t ovf = 0;
	HIPCHK(hipMemcpyAsync(&ovf, overflow, 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(g_last_bf_clocks, a.clocks, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	(void) hipEventElapsedTime(&g_last_bf_gemm_ms, e0, e1);
	if (ovf)      // a candidate list overflowed (bound far too loose for some query): canonical scan instead
		return hnsw_gpu_bruteforce_dev(ix, d_queries, nq, k, d_idx, d_dists, stream_);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_bruteforce_tile(void) { return g_last_bf_tile; }

/* device time of the MFMA filter kernel of the most recent hnsw_gpu_bruteforce_mfma_dev call */
extern "C" float hnsw_gpu_last_bruteforce_gemm_ms(void) { return g_last_bf_gemm_ms; }

/* shader-clock MHz during that kernel: ticks of the shader clock over ticks of the constant 100 MHz clock, both taken by block 0
 * around its K loop — what the matrix roof has to be priced at when the device does not hold its nominal clock under this load */
extern "C" double hnsw_gpu_last_bruteforce_clock_mhz(void)
{
	int khz = 0, dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
		khz = 100000;
	return g_last_bf_clocks[1] ? khz * 1e-3 * (double) g_last_bf_clocks[0] / (double) g_last_bf_clocks[1] : 0.0;
}

// ------------------------------------------------------------------------------------
// multi-shard merge: nlists x (dist,label) lists per query -> ef best by (dist, label)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool dl_less(uint32_t da, uint64_t la, uint32_t db, uint64_t lb)
{
	return da < db || (da == db && la < lb);
}

__global__ __launch_bounds__(64) void topk_merge_kernel(const uint64_t *__restrict__ in_labels, const float *__restrict__ in_dists,
														size_t lstep, size_t dstep,       /* list-to-list strides, in elements */
														uint32_t nlists, uint32_t nq, uint32_t ef,
														uint64_t *__restrict__ out_labels, float *__restrict__ out_dists,
														uint32_t *__restrict__ out_counts)
{
	const uint32_t qi = blockIdx.x;
	const int lane = threadIdx.x;
	const uint32_t total = nlists * ef;
	uint32_t kept = 0;
	for (uint32_t x0 = 0; x0 < total; x0 += 64)
	{
		const uint32_t x = x0 + lane;
		bool emit = false;
		if (x < total)
		{
			const uint32_t l = x / ef, i = x - l * ef;
			const size_t at = (size_t) qi * ef + i;
			const uint64_t lab = in_labels[l * lstep + at];
			const uint32_t d = ord_f32(in_dists[l * dstep + at]);
			if (lab != ~0ull)
			{
				uint32_t rank = i;
				for (uint32_t m = 0; m < nlists && rank < ef; m++)
				{
					if (m == l) continue;
					const size_t ob = (size_t) qi * ef;
					uint32_t lo = 0, hi = ef;
					while (lo < hi)
					{
						const uint32_t mid = (lo + hi) >> 1;
						const uint64_t ol = in_labels[m * lstep + ob + mid];
						const uint32_t od = ord_f32(in_dists[m * dstep + ob + mid]);
						// equal keys (cannot happen for disjoint shards) go to the lower list number
						const bool below = (ol != ~0ull) && (dl_less(od, ol, d, lab) || (od == d && ol == lab && m < l));
						if (below) lo = mid + 1; else hi = mid;
					}
					rank += lo;
				}
				if (rank < ef)
				{
					out_labels[(size_t) qi * ef + rank] = lab;
					if (out_dists) out_dists[(size_t) qi * ef + rank] = unord_f32(d);
					emit = true;
				}
			}
		}
		kept += (uint32_t) __builtin_popcountll(__ballot(emit));
	}
	for (uint32_t i = kept + lane; i < ef; i += 64)
	{
		out_labels[(size_t) qi * ef + i] = ~0ull;
		if (out_dists) out_dists[(size_t) qi * ef + i] = __builtin_inff();
	}
	if (lane == 0) out_counts[qi] = kept;
}

extern "C" int hnsw_gpu_merge_topk_strided_dev(int device, const label_t *d_in_labels, size_t label_list_stride,
											   const dist_t *d_in_dists, size_t dist_list_stride, size_t nlists,
											   size_t nq, size_t ef, label_t *d_out_labels, dist_t *d_out_dists,
											   uint32_t *d_out_counts, void *stream)
{
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_in_labels || !d_in_dists || !d_out_labels || !d_out_counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (nlists == 0 || ef == 0) return fail(HNSW_GPU_ERR_ARG, "nlists and ef must be positive");
	if (nlists * ef >= 0xFFFFFFFFull || nq >= 0x7FFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "merge too large");
	if (label_list_stride < nq * ef || dist_list_stride < nq * ef) return fail(HNSW_GPU_ERR_ARG, "list stride smaller than one list");
	HIPCHK(hipSetDevice(device));
	hipLaunchKernelGGL(topk_merge_kernel, dim3((uint32_t) nq), dim3(64), 0, (hipStream_t) stream, d_in_labels, d_in_dists,
					   label_list_stride, dist_list_stride, (uint32_t) nlists, (uint32_t) nq, (uint32_t) ef, d_out_labels, d_out_dists,
					   d_out_counts);
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_merge_topk_dev(int device, const label_t *d_in_labels, const dist_t *d_in_dists, size_t nlists,
									   size_t nq, size_t ef, label_t *d_out_labels, dist_t *d_out_dists,
									   uint32_t *d_out_counts, void *stream)
{
	return hnsw_gpu_merge_topk_strided_dev(device, d_in_labels, nq * ef, d_in_dists, nq * ef, nlists, nq, ef, d_out_labels,
										   d_out_dists, d_out_counts, stream);
}

// ------------------------------------------------------------------------------------
// insert path: link stored elements into the graph (device_build.h)
// ------------------------------------------------------------------------------------
extern "C" int pgemb_sort_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, int n, void *stream);

typedef void (*build_kernel_t)(const BuildArgs);

// ext_*: the candidates of ONE new element (count == 1) as a search already produced them — ascending by (dist, idx), the order
// searchBaseLayer's results leave hnsw_gpu_search_base* in — in memory the device can read (pinned host memory will do): the link
// step then runs without a search of its own (hnsw_gpu_index_insert_candidates).
static int link_range(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio, void *stream_,
					  const uint32_t *ext_idx, const float *ext_dist, const uint32_t *ext_cnt)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (first + count > ix->n) return fail(HNSW_GPU_ERR_ARG, "elements [%zu, %zu) are not stored (count %zu)", first, first + count, ix->n);
	if (count == 0) return HNSW_GPU_OK;
	if (ext_idx && (count != 1 || !ext_dist || !ext_cnt)) return fail(HNSW_GPU_ERR_ARG, "external candidates are for one element");
	if (max_batch == 0) max_batch = 4096;
	if (ratio == 0) ratio = 8;
	const size_t efc = ix->meta.efConstruction, M = ix->meta.M, maxM = ix->meta.maxM;
	if (efc == 0 || M == 0 || M > maxM) return fail(HNSW_GPU_ERR_ARG, "bad efConstruction/M");
	HIPCHK(hipSetDevice(ix->device));
	hipStream_t stream = (hipStream_t) stream_;

	// scratch carve
	max_batch = std::min(max_batch, count);
	const size_t slots = max_batch * M;
	if (slots >= 0x7FFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "batch too large");
	size_t tmp_bytes = 0;
	if (pgemb_sort_u64(nullptr, &tmp_bytes, nullptr, nullptr, (int) slots, stream) != 0)
		return fail(HNSW_GPU_ERR_HIP, "radix sort sizing failed");
	const size_t o_idx = 0;
	const size_t o_dist = o_idx + round_up(max_batch * efc * 4, 256);
	const size_t o_cnt = o_dist + round_up(max_batch * efc * 4, 256);
	const size_t o_pairs = o_cnt + round_up(max_batch * 4, 256);
	const size_t o_sorted = o_pairs + round_up(slots * 8, 256);
	const size_t o_seg = o_sorted + round_up(slots * 8, 256);
	const size_t o_ctr = o_seg + round_up(slots * 4, 256);
	const size_t o_tmp = o_ctr + 256;
	const size_t total = o_tmp + round_up(tmp_bytes, 256);
	if (max_batch > ix->bld_batch || tmp_bytes > ix->bld_tmp_bytes || !ix->bld)
	{
		if (ix->bld) (void) hipFree(ix->bld);
		ix->bld = nullptr; ix->bld_batch = 0;
		HIPCHK(hipMalloc(&ix->bld, total));
		ix->bld_batch = max_batch; ix->bld_tmp_bytes = tmp_bytes;
	}
	char *B = (char *) ix->bld;
	uint32_t *cand_idx = (uint32_t *) (B + o_idx);
	float *cand_dist = (float *) (B + o_dist);
	uint32_t *cand_cnt = (uint32_t *) (B + o_cnt);
	uint64_t *pairs = (uint64_t *) (B + o_pairs), *sorted = (uint64_t *) (B + o_sorted);
	uint32_t *seg = (uint32_t *) (B + o_seg), *ctr = (uint32_t *) (B + o_ctr);

	BuildArgs a;
	memset(&a, 0, sizeof(a));
	a.vec = ix->vec; a.links = ix->links;
	a.dim = (uint32_t) ix->meta.dim; a.stride = ix->stride; a.nchunks = ix->stride / 4; a.kiters = (a.nchunks + 15) / 16;
	a.qpad_floats = (uint32_t) round_up(a.kiters, BUILD_KB) * 64;
	a.maxM = (uint32_t) maxM; a.M = (uint32_t) M; a.lstride = ix->lstride; a.efc = (uint32_t) efc;
	a.cand_idx = cand_idx; a.cand_dist = cand_dist; a.cand_cnt = cand_cnt;
	a.pairs = pairs; a.npairs = ctr; a.sorted_pairs = sorted; a.seg_start = seg; a.nseg = ctr + 1; a.ticket = ctr + 2;
	const uint32_t cap = (uint32_t) round_up(std::max<size_t>(std::max(efc, maxM + 1), 128), 8);   // tmpd holds 2 x 64 sums
	a.wave_bytes = (uint32_t) round_up((size_t) a.qpad_floats * 4 + (size_t) cap * (8 * 2 + 4 * 3) + (maxM + 2) * 4, 16);
	if (a.wave_bytes > LDS_PER_CU) return fail(HNSW_GPU_ERR_ARG, "efConstruction/maxM/dim need too much LDS (%u bytes)", a.wave_bytes);
	uint32_t wpb = 4;
	while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1;
	const size_t lds = (size_t) wpb * a.wave_bytes;
	build_kernel_t ksel, krev;
	switch ((int) ix->meta.dist_func)
	{
		case F_L2:     ksel = select_links_kernel<F_L2>;        krev = reverse_links_kernel<F_L2>; break;
		case F_COSINE: ksel = select_links_kernel<F_COSINE>;    krev = reverse_links_kernel<F_COSINE>; break;
		default:       ksel = select_links_kernel<F_MANHATTAN>; krev = reverse_links_kernel<F_MANHATTAN>; break;
	}
	if (lds > 48 * 1024)
	{
		HIPCHK(hipFuncSetAttribute((const void *) ksel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		HIPCHK(hipFuncSetAttribute((const void *) krev, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	}

	size_t linked = first, end = first + count;
	if (linked == 0) linked = 1;                    // element 0 is never bound: hnswalg.cpp:228
	while (linked < end)
	{
		const size_t b = std::min({end - linked, max_batch, std::max<size_t>(1, linked / ratio)});
		const size_t bslots = b * M;
		const bool single = b == 1;                     // the reference's serial insert: no sort, no segment marking (device_build.h)
		HIPCHK(hipMemsetAsync(ctr, 0, 16, stream));
		if (!single) HIPCHK(hipMemsetAsync(pairs, 0xFF, bslots * 8, stream));
		// 1. searchBaseLayer(ef = efConstruction) for every new element (hnswalg.cpp:229) — unless the caller brought its result
		if (ext_idx)
		{
			a.cand_idx = ext_idx; a.cand_dist = ext_dist; a.cand_cnt = ext_cnt;
		}
		else
		{
			a.cand_idx = cand_idx; a.cand_dist = cand_dist; a.cand_cnt = cand_cnt;
			int rc = launch_search(ix, &ix->ws, ix->vec + linked * ix->stride, ix->stride, b, efc, 1, nullptr, cand_idx, cand_dist,
								   cand_cnt, nullptr, stream);
			if (rc) return rc;
		}
		// 2. choose links, emit reverse pairs
		a.first = (uint32_t) linked; a.count = (uint32_t) b; a.pair_slots = (uint32_t) bslots;
		a.single = single ? 1u : 0u; a.seg_out = seg; a.nseg_out = ctr + 1;
		a.sorted_pairs = single ? pairs : sorted;
		hipLaunchKernelGGL(ksel, dim3((uint32_t) ((b + wpb - 1) / wpb)), dim3(wpb * 64), lds, stream, a);
		// 3. reverse edges grouped by target
		if (!single)
		{
			size_t tb = ix->bld_tmp_bytes;
			if (pgemb_sort_u64(B + o_tmp, &tb, pairs, sorted, (int) bslots, stream) != 0)
				return fail(HNSW_GPU_ERR_HIP, "radix sort failed");
			hipLaunchKernelGGL(mark_segments_kernel, dim3((uint32_t) ((bslots + 255) / 256)), dim3(256), 0, stream, sorted,
							   (uint32_t) bslots, seg, ctr + 1);
		}
		const uint32_t rblocks = (uint32_t) std::min<size_t>((bslots + wpb - 1) / wpb, (size_t) ix->num_cu * 4);
		hipLaunchKernelGGL(krev, dim3(rblocks), dim3(wpb * 64), lds, stream, a);
		HIPCHK(hipGetLastError());
		linked += b;
	}
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_link(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio,
								   void *stream_)
{
	return link_range(ix, first, count, max_batch, ratio, stream_, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------
// small accessors used by the drop-in insert (embedding_shim.cpp)
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_index_reserve(hnsw_gpu_index *ix, size_t capacity)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (capacity <= ix->cap) return HNSW_GPU_OK;
	if (capacity >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "capacity exceeds idx_t range");
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipDeviceSynchronize());
	float *nv = nullptr; uint32_t *nl = nullptr; uint64_t *nb = nullptr;
	hipError_t e;
	if ((e = hipMalloc(&nv, capacity * ix->stride * sizeof(float))) != hipSuccess ||
		(e = hipMalloc(&nl, capacity * ix->lstride * sizeof(uint32_t))) != hipSuccess ||
		(e = hipMalloc(&nb, capacity * sizeof(uint64_t))) != hipSuccess)
	{
		if (nv) (void) hipFree(nv);
		if (nl) (void) hipFree(nl);
		if (nb) (void) hipFree(nb);
		return fail(HNSW_GPU_ERR_NOMEM, "cannot grow the mirror to %zu elements: %s", capacity, hipGetErrorString(e));
	}
	HIPCHK(hipMemcpy(nv, ix->vec, ix->n * ix->stride * sizeof(float), hipMemcpyDeviceToDevice));
	HIPCHK(hipMemcpy(nl, ix->links, ix->n * ix->lstride * sizeof(uint32_t), hipMemcpyDeviceToDevice));
	HIPCHK(hipMemcpy(nb, ix->labels, ix->n * sizeof(uint64_t), hipMemcpyDeviceToDevice));
	(void) hipFree(ix->vec); (void) hipFree(ix->links); (void) hipFree(ix->labels);
	ix->vec = nv; ix->links = nl; ix->labels = nb;
	ix->cap = capacity;
	// the visited bitmaps are sized by capacity: drop them, the next search re-creates them
	if (ix->ws.vis) (void) hipFree(ix->ws.vis);
	if (ix->ws.vlog) (void) hipFree(ix->ws.vlog);
	ix->ws.vis = nullptr; ix->ws.vlog = nullptr; ix->ws.vis_slots = 0; ix->ws.vis_words = 0;
	ix->generation++;         // contexts notice and rebuild their bitmaps
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_get_links(hnsw_gpu_index *ix, idx_t idx, idx_t *out)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !out || idx >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx);
	HIPCHK(hipSetDevice(ix->device));
	const size_t maxM = ix->meta.maxM;
	uint32_t tmp[4096 + 16];
	HIPCHK(hipMemcpy(tmp, ix->links + (size_t) idx * ix->lstride, ix->lstride * 4, hipMemcpyDeviceToHost));
	uint32_t cnt = 0;
	for (size_t j = 0; j < maxM; j++)
		if (tmp[j] != LINK_NONE) out[1 + cnt++] = tmp[j];
	out[0] = cnt;
	for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	return HNSW_GPU_OK;
}

// The link list of one element and the lists of all its neighbours in one launch + one wait: what an insert changed
// (hnswalg.cpp:169-222: the new element's list and a reverse link in each neighbour's), for the write-back of
// hnsw_bind_point.  Rows land in the mirror's pinned staging; block 0 = the element, block 1+j = its j-th link slot.
// done_ctr / flag (hnsw_gpu_index_insert_*): the block that finishes LAST stores the completion flag behind a system-scope release —
// the lists of every block are in host memory before the flag, and no extra launch is needed for it.
__global__ __launch_bounds__(64) void gather_link_lists_kernel(const uint32_t *__restrict__ links, uint32_t lstride, uint32_t idx,
															   uint32_t n, uint32_t *__restrict__ out, uint32_t *done_ctr, uint32_t *flag)
{
	uint32_t src = idx;
	bool have = true;
	if (blockIdx.x > 0)
	{
		src = links[(size_t) idx * lstride + (blockIdx.x - 1)];
		have = src != LINK_NONE && src < n;
	}
	for (uint32_t j = threadIdx.x; j < lstride; j += 64)
		out[(size_t) blockIdx.x * lstride + j] = have ? links[(size_t) src * lstride + j] : LINK_NONE;
	if (flag)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: this block's rows
		uint32_t last = 0;
		if (threadIdx.x == 0) last = atomicAdd(done_ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
		if (__builtin_amdgcn_readfirstlane(last))
		{
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");    // every other block's release happened before its increment
			if (threadIdx.x == 0)
			{
				atomicExch(done_ctr, 0u);                     // ready for the next insert: no memset between calls
				__hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
}

extern "C" int hnsw_gpu_index_get_link_lists(hnsw_gpu_index *ix, idx_t idx, idx_t *mine, idx_t *others)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !mine || !others || idx >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx);
	HIPCHK(hipSetDevice(ix->device));
	const size_t maxM = ix->meta.maxM, ls = ix->lstride;
	const size_t need = (maxM + 1) * ls * 4;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, std::max<size_t>(need, 64 << 10), hipHostMallocDefault));
		ix->pin_bytes = std::max<size_t>(need, 64 << 10);
	}
	uint32_t *h = (uint32_t *) ix->pin;
	hipLaunchKernelGGL(gather_link_lists_kernel, dim3((uint32_t) maxM + 1), dim3(64), 0, 0, ix->links, (uint32_t) ls, (uint32_t) idx,
					   (uint32_t) ix->n, h, (uint32_t *) nullptr, (uint32_t *) nullptr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(nullptr));
	auto compact = [&](const uint32_t *row, idx_t *out)
	{
		uint32_t cnt = 0;
		for (size_t j = 0; j < maxM; j++)
			if (row[j] != LINK_NONE) out[1 + cnt++] = row[j];
		out[0] = cnt;
		for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	};
	compact(h, mine);
	// neighbour j of the compacted list sits in link slot s_j of the row (slots may hold holes)
	size_t k = 0;
	for (size_t s = 0; s < maxM && k < mine[0]; s++)
		if (h[s] != LINK_NONE) { compact(h + (1 + s) * ls, others + k * (maxM + 1)); k++; }
	return HNSW_GPU_OK;
}

// Gathered link rows (row 0 = the element, row 1 + s = the element in its link slot s) -> the compacted [count | links] lists of
// the element and of each neighbour, in the order of the compacted list.
static void compact_lists(const uint32_t *lists, size_t maxM, size_t ls, idx_t *mine, idx_t *others)
{
	auto compact = [&](const uint32_t *row, idx_t *out)
	{
		uint32_t cnt = 0;
		for (size_t j = 0; j < maxM; j++)
			if (row[j] != LINK_NONE) out[1 + cnt++] = row[j];
		out[0] = cnt;
		for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	};
	compact(lists, mine);
	size_t k = 0;
	for (size_t s2 = 0; s2 < maxM && k < mine[0]; s2++)
		if (lists[s2] != LINK_NONE) { compact(lists + (1 + s2) * ls, others + k * (maxM + 1)); k++; }
}

typedef void (*insert_kernel_t)(const InsertArgs);

static std::atomic<uint64_t> g_inserts_two_launch{0}, g_inserts_general{0};
extern "C" void hnsw_gpu_insert_path_counts(uint64_t out[2])
{
	if (!out) return;
	out[0] = g_inserts_two_launch.load();
	out[1] = g_inserts_general.load();
}

static bool insert_fused_wanted()
{
	knobs_init();
	return knob(K_INSERT_FUSED, 1) != 0;                     // 0: the four-launch path of round 3's first half (A/B runs, tests of both)
}

// The two-launch insert's shape for this mirror: *lds = 0 when max(efConstruction, maxM + 1) candidates are more than the chain
// of device_insert.h keeps in one wavefront's registers (INS_MAX_SIDE) or a block does not fit a CU's LDS — the caller then takes
// the general builder path.  Device scratch (ix->ins): candidates of the insert's own walk | targets | bit matrix.
static int plan_insert(hnsw_gpu_index *ix, InsertArgs *a, size_t *lds)
{
	*lds = 0;
	memset(a, 0, sizeof(*a));
	const size_t efc = ix->meta.efConstruction, M = ix->meta.M, maxM = ix->meta.maxM;
	if (efc == 0 || M == 0 || M > maxM) return fail(HNSW_GPU_ERR_ARG, "bad efConstruction/M");
	BuildArgs &b = a->b;
	b.vec = ix->vec; b.links = ix->links;
	b.dim = (uint32_t) ix->meta.dim; b.stride = ix->stride; b.nchunks = ix->stride / 4; b.kiters = (b.nchunks + 15) / 16;
	b.qpad_floats = (uint32_t) round_up(b.kiters, INS_KB) * 64;
	b.maxM = (uint32_t) maxM; b.M = (uint32_t) M; b.lstride = ix->lstride; b.efc = (uint32_t) efc;
	const size_t side = round_up(std::max(efc, maxM + 1), 64);
	if (side > INS_MAX_SIDE) return HNSW_GPU_OK;
	const size_t cap = side;
	const size_t shared = cap * 8 * 3 + side * (side / 64) * 8 + round_up(maxM + 2, 4) * 4 + 16;
	const size_t per_wave = ((size_t) b.qpad_floats + 128) * 4;
	const size_t lds_limit = std::min(LDS_PER_CU, ix->max_lds);                     // what ONE block may ask for on this device
	if (lds_limit < 2048) return HNSW_GPU_OK;
	size_t nw = 8;
	while (nw > 1 && shared + nw * per_wave > lds_limit - 1024) nw >>= 1;
	if (shared + nw * per_wave > lds_limit - 1024) return HNSW_GPU_OK;
	size_t nw2 = 12;                                                                // step 2: up to 12 wavefronts around one target (device_insert.h)
	while (nw2 > nw && shared + nw2 * per_wave > lds_limit - 1024) nw2 -= 4;
	if (nw2 < nw) nw2 = nw;
	a->nw = (uint32_t) nw; a->nw2 = (uint32_t) nw2; a->side = (uint32_t) side; a->cap = (uint32_t) cap;
	const size_t o_ci = 0, o_cd = o_ci + round_up(efc * 4, 256), o_cc = o_cd + round_up(efc * 4, 256);
	const size_t o_tg = o_cc + 256, o_bits = o_tg + round_up(M * 4, 256), total = o_bits + side * (side / 16) * 2;
	if (ix->ins_bytes < total)
	{
		if (ix->ins) (void) hipFree(ix->ins);
		ix->ins = nullptr; ix->ins_bytes = 0;
		HIPCHK(hipMalloc(&ix->ins, total));
		ix->ins_bytes = total;
	}
	char *S = (char *) ix->ins;
	b.cand_idx = (const uint32_t *) (S + o_ci); b.cand_dist = (const float *) (S + o_cd); b.cand_cnt = (const uint32_t *) (S + o_cc);
	a->targets = (uint32_t *) (S + o_tg); a->bits = (uint16_t *) (S + o_bits);
	a->labels = ix->labels;
	a->ntargets = ix->misc + 5; a->done1 = ix->misc + 4; a->done2 = ix->misc + 3;
	*lds = shared + nw2 * per_wave;                                                 // (the larger of the two carves: one attribute for both kernels)
	return HNSW_GPU_OK;
}

// hnsw_bind_point's device side in ONE host call (hnswalg.cpp:279-291, 225-232): element `idx` (= the mirror's current count)
// is appended and linked exactly as the reference's serial insert links it, and the changed link lists — its own and one per
// selected neighbour — come back compacted ([count | links], maxM + 1 words each) for the host's write-back.  Everything is
// enqueued on the default stream without a host wait in between: the row and its label are read by the append kernel straight
// from pinned host memory, the gathered lists are written straight into it, and a one-thread kernel behind them stores a
// completion flag that the calling core polls (no copy engine, no interrupt wake-up: the few-queries mechanics of
// hnsw_gpu_search_batch).  Round 2 made this call as append (2 blocking copies + sync) + link (2 memsets, search, select, a
// hipCUB radix sort, segment marking, reverse) + get_link_lists (launch + sync): 0.85-1.5 ms per row against the reference's
// 0.06-0.12 ms; a single row needs no sort (its neighbours are distinct targets) and no waits.
static int insert_impl(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx, const dist_t *cand_dist,
					   uint32_t ncand, idx_t *mine, idx_t *others)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !point || !mine || !others) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (cand_idx && (!cand_dist || ncand > ix->meta.efConstruction)) return fail(HNSW_GPU_ERR_ARG, "bad candidate list");
	if (cand_idx)
	{
		// the kernels use these numbers as row and link addresses and rely on searchBaseLayer's order: stored elements only,
		// strictly ascending by (dist, idx) — which also makes them distinct (at most 512 entries: nothing next to the insert)
		for (uint32_t i = 0; i < ncand; i++)
		{
			if (cand_idx[i] >= idx) return fail(HNSW_GPU_ERR_ARG, "candidate %u is element %u, not below the new element %u", i, (unsigned) cand_idx[i], (unsigned) idx);
			if (cand_dist[i] != cand_dist[i]) return fail(HNSW_GPU_ERR_ARG, "candidate %u has a NaN distance", i);
			if (i > 0 && !(cand_dist[i - 1] < cand_dist[i] || (cand_dist[i - 1] == cand_dist[i] && cand_idx[i - 1] < cand_idx[i])))
				return fail(HNSW_GPU_ERR_ARG, "candidates %u and %u are not in ascending (dist, idx) order", i - 1, i);
		}
	}
	if ((size_t) idx != ix->n) return fail(HNSW_GPU_ERR_ARG, "insert_one(%u): the mirror holds %zu elements", (unsigned) idx, ix->n);
	if (ix->n + 1 > ix->cap) return fail(HNSW_GPU_ERR_ARG, "insert exceeds capacity (%zu)", ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes the staging
	if (ix->ins_dirty)
	{
		// a previous insert failed after its kernels were enqueued: the last-block counters (misc words 3..5) may be non-zero, and
		// with them every later insert would mis-detect its last block and never store the completion flag
		HIPCHK(hipDeviceSynchronize());
		HIPCHK(hipMemset(ix->misc + 3, 0, 12));
		ix->ins_dirty = false;
	}
	const size_t dim = ix->meta.dim, maxM = ix->meta.maxM, ls = ix->lstride;
	const size_t efc_ = ix->meta.efConstruction;
	const size_t o_lab = round_up(dim * 4, 8), o_lists = round_up(o_lab + 8, 256), o_flag = o_lists + round_up((maxM + 1) * ls * 4, 256);
	const size_t o_ci = o_flag + 256, o_cd = o_ci + round_up(efc_ * 4, 256), o_cc = o_cd + round_up(efc_ * 4, 256);
	const size_t need = o_cc + 256;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, std::max<size_t>(need, 64 << 10), hipHostMallocDefault));
		ix->pin_bytes = std::max<size_t>(need, 64 << 10);
	}
	char *h = ix->pin;
	memcpy(h, point, dim * 4);
	memcpy(h + o_lab, &label, 8);
	volatile uint32_t *flag = (volatile uint32_t *) (h + o_flag);
	*flag = 0;
	uint32_t *lists = (uint32_t *) (h + o_lists);
	int rc;
	InsertArgs ia;
	size_t ilds = 0;
	insert_kernel_t ksel = nullptr, krev = nullptr;
	bool two_launch = insert_fused_wanted() && plan_insert(ix, &ia, &ilds) == HNSW_GPU_OK && ilds;
	if (two_launch)
	{
		switch ((int) ix->meta.dist_func)
		{
			case F_L2:     ksel = insert_select_kernel<F_L2>;        krev = insert_reverse_kernel<F_L2>; break;
			case F_COSINE: ksel = insert_select_kernel<F_COSINE>;    krev = insert_reverse_kernel<F_COSINE>; break;
			default:       ksel = insert_select_kernel<F_MANHATTAN>; krev = insert_reverse_kernel<F_MANHATTAN>; break;
		}
		static std::atomic<size_t> lds_allowed[3][8];          // per function and device: the attribute is set when a larger carve comes along, not per insert
		std::atomic<size_t> &allowed = lds_allowed[std::min(std::max((int) ix->meta.dist_func, 0), 2)][ix->device & 7];
		if (ilds > 48 * 1024 && (ilds > allowed.load() || ix->device > 7))
		{
			// a device that refuses the carve takes the general builder path (plan_insert's contract), it does not fail the insert
			if (hipFuncSetAttribute((const void *) ksel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ilds) != hipSuccess ||
				hipFuncSetAttribute((const void *) krev, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ilds) != hipSuccess)
			{
				(void) hipGetLastError();
				two_launch = false;
			}
			else allowed.store(ilds);
		}
	}
	if (two_launch)
	{
		// two launches (device_insert.h): [append +] pair triangle + chain | one block per target + the flag
		ia.b.first = (uint32_t) idx; ia.b.count = 1;
		ia.bind = idx > 0 ? 1u : 0u;
		ia.lists_out = lists; ia.flag = (uint32_t *) (h + o_flag);
		if (cand_idx || idx == 0)                            // the walk has been done (and validated) already: the kernel reads its result, the row and the label from pinned memory
		{
			if (cand_idx)
			{
				memcpy(h + o_ci, cand_idx, (size_t) ncand * 4);
				memcpy(h + o_cd, cand_dist, (size_t) ncand * 4);
			}
			*(uint32_t *) (h + o_cc) = ncand;
			ia.ncand_p1 = ncand + 1;                         // known here: the kernels do not fetch it over the bus
			ia.b.cand_idx = (const uint32_t *) (h + o_ci); ia.b.cand_dist = (const float *) (h + o_cd); ia.b.cand_cnt = (const uint32_t *) (h + o_cc);
			ia.src_row = (const float *) h; ia.src_label = (const uint64_t *) (h + o_lab);
			ix->n += 1;                                      // stored by step 1's block 0
		}
		else                                                 // searchBaseLayer(ef = efConstruction) of the insert itself (hnswalg.cpp:229), the point read as
		{                                                    // the query straight from pinned memory over the idx elements stored so far; step 1 stores the row
			rc = launch_search(ix, &ix->ws, (const float *) h, dim, 1, efc_, 1, nullptr, (uint32_t *) ia.b.cand_idx,
							   (float *) ia.b.cand_dist, (uint32_t *) ia.b.cand_cnt, nullptr, nullptr);
			if (rc) return rc;
			ia.src_row = (const float *) h; ia.src_label = (const uint64_t *) (h + o_lab);
			ix->n += 1;
		}
		// (one wavefront per unit of the bit triangle: device_insert.h)
		const uint32_t g1 = std::max<uint32_t>(1u, (units_for((uint32_t) efc_) + ia.nw - 1) / ia.nw);
		hipLaunchKernelGGL(ksel, dim3(g1), dim3(ia.nw * 64), ilds, 0, ia);
		if (hipError_t le = hipGetLastError(); le != hipSuccess)
		{
			ix->n = idx;                                     // nothing was stored
			return fail(HNSW_GPU_ERR_HIP, "insert step 1 did not launch: %s", hipGetErrorString(le));
		}
		hipLaunchKernelGGL(krev, dim3((uint32_t) ix->meta.M), dim3(ia.nw2 * 64), ilds, 0, ia);
		if (hipError_t le = hipGetLastError(); le != hipSuccess)
		{
			// step 1 runs (row, label, own list), step 2 never will: no element points at the new one, so the mirror without it is
			// the mirror before the call; step 1's last block has reset its own counter, the next insert re-checks all three
			(void) hipStreamSynchronize(nullptr);
			ix->n = idx;
			ix->ins_dirty = true;
			return fail(HNSW_GPU_ERR_HIP, "insert step 2 did not launch: %s", hipGetErrorString(le));
		}
		rc = poll_done_flag(flag, "an insert", nullptr);
		if (rc)
		{
			// the element is stored and (partly) linked: the graph is searchable but not the reference's; the caller sees the error
			// and re-mirrors (embedding_shim.cpp drops its mirror on any insert error)
			ix->ins_dirty = true;
			return rc;
		}
		compact_lists(lists, maxM, ls, mine, others);
		g_inserts_two_launch++;
		return HNSW_GPU_OK;
	}
	g_inserts_general++;
	rc = hnsw_gpu_index_append_dev(ix, (const coord_t *) h, (const label_t *) (h + o_lab), 1, nullptr);
	if (rc) return rc;
	if (idx > 0)                                             // element 0 is never bound (hnswalg.cpp:228)
	{
		if (cand_idx)                                        // the walk has been done (and validated) already: its result, from pinned memory
		{
			memcpy(h + o_ci, cand_idx, (size_t) ncand * 4);
			memcpy(h + o_cd, cand_dist, (size_t) ncand * 4);
			*(uint32_t *) (h + o_cc) = ncand;
			rc = link_range(ix, idx, 1, 1, 0, nullptr, (const uint32_t *) (h + o_ci), (const float *) (h + o_cd), (const uint32_t *) (h + o_cc));
		}
		else
			rc = link_range(ix, idx, 1, 1, 0, nullptr, nullptr, nullptr, nullptr);
		if (rc) return rc;
	}
	// (the gather's last block stores the flag; its block counter is word 3 of the mirror's misc words: zero at creation, reset by that block)
	hipLaunchKernelGGL(gather_link_lists_kernel, dim3((uint32_t) maxM + 1), dim3(64), 0, 0, ix->links, (uint32_t) ls, (uint32_t) idx,
					   (uint32_t) ix->n, lists, ix->misc + 3, (uint32_t *) (h + o_flag));
	HIPCHK(hipGetLastError());
	rc = poll_done_flag(flag, "an insert", nullptr);
	if (rc) { ix->ins_dirty = true; return rc; }
	compact_lists(lists, maxM, ls, mine, others);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_insert_one(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, idx_t *mine, idx_t *others)
{
	return insert_impl(ix, point, label, idx, nullptr, nullptr, 0, mine, others);
}

// The same with the candidate list given: what searchBaseLayer(point, ef = efConstruction) returned on THIS mirror a moment ago
// (hnsw_gpu_search_trace in base mode: element numbers and distances ascending by (dist, idx)) — a caller that has just walked for
// the point (the validated cache of the unmodified glue walks to CHECK the mirror, shim_cache.h) does not pay for the walk twice.
extern "C" int hnsw_gpu_index_insert_candidates(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx,
												const dist_t *cand_dist, uint32_t ncand, idx_t *mine, idx_t *others)
{
	if (!cand_idx || !cand_dist) return fail(HNSW_GPU_ERR_ARG, "NULL candidate list");
	return insert_impl(ix, point, label, idx, cand_idx, cand_dist, ncand, mine, others);
}

// Refresh part of the mirror from the host: element images of [first, first+count) replace what
// the mirror holds (links, vector, label); elements past the current end are added.  This is the
// incremental counterpart of create_from_flat for a host that tracks which pages changed
// (new elements, re-linked neighbours, vacuum flags) — SURVEY.md §8(f) rank 2.
extern "C" int hnsw_gpu_index_update_from_flat(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count)
{
	if (!ix || (count && !elements)) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::unique_lock<std::recursive_mutex> lock_(ix->mu);
	if (first > ix->n) return fail(HNSW_GPU_ERR_ARG, "update would leave a gap (first %zu > count %zu)", first, ix->n);
	HIPCHK(hipSetDevice(ix->device));
	const size_t end = first + count;
	if (end > ix->cap)
	{
		int rc = hnsw_gpu_index_reserve(ix, end + end / 2);
		if (rc) return rc;
	}
	const size_t n_total = std::max(ix->n, end);
	int rc = import_range(ix, elements, first, count, n_total);
	if (rc) return rc;
	ix->n = n_total;
	ix->xnorm_n = 0;            // cached row norms are stale
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// search contexts: independent batches in flight on different streams
// ------------------------------------------------------------------------------------
struct hnsw_gpu_ctx
{
	hnsw_gpu_index *ix;
	SearchWs ws;
	// host-pointer form (hnsw_gpu_search_batch_ctx_host): own stream + device staging, grow-only
	hipStream_t stream = nullptr;
	void *stage = nullptr; size_t stage_bytes = 0;
};

extern "C" int hnsw_gpu_ctx_create(hnsw_gpu_index *ix, hnsw_gpu_ctx **out)
{
	if (!ix || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(ix->device));
	hnsw_gpu_ctx *c = new (std::nothrow) hnsw_gpu_ctx();
	if (!c) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	c->ix = ix;
	int rc = ws_init(&c->ws);
	if (rc) { ws_free(&c->ws); delete c; return rc; }
	*out = c;
	return HNSW_GPU_OK;
}

extern "C" void hnsw_gpu_ctx_destroy(hnsw_gpu_ctx *c)
{
	if (!c) return;
	(void) hipSetDevice(c->ix->device);
	ws_free(&c->ws);
	if (c->stage) (void) hipFree(c->stage);
	if (c->stream) (void) hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int hnsw_gpu_search_batch_ctx(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef,
										 label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										 void *stream)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	return launch_search(c->ix, &c->ws, d_queries, c->ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats,
						 (hipStream_t) stream);
}

// 8-wave team blocks the device holds at once for rows wider than 320 floats (one per CU: 2 waves/SIMD): the figure a host sizes
// hnsw_gpu_ctx_set_walkers by.  <= 0: no such device.
extern "C" int hnsw_gpu_device_blocks(int device)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void) hipGetLastError(); return 0; }
	return prop.multiProcessorCount;
}

extern "C" int hnsw_gpu_ctx_set_walkers(hnsw_gpu_ctx *c, unsigned per_block)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	c->ws.walkers_hint = per_block;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_ctx_search_ms(hnsw_gpu_ctx *c, unsigned back, float *ms)
{
	if (!c || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return ws_search_ms(c->ix->device, &c->ws, back, ms);
}

// Host-pointer form of a context search: copy in, launch, copy out on the context's own stream and
// wait for that stream only, so host threads that own one context each keep several batches in
// flight on the device (the batching server's dispatchers, server_main.cpp).  Buffers from
// hnsw_gpu_host_alloc make the copies true DMA transfers.  One caller at a time per context.
extern "C" int hnsw_gpu_search_batch_ctx_host(hnsw_gpu_ctx *c, const coord_t *queries, size_t nq, size_t ef,
											  label_t *labels, dist_t *dists, uint32_t *counts)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	hnsw_gpu_index *ix = c->ix;
	HIPCHK(hipSetDevice(ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	const size_t dim = ix->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	if (qb + lb + db + cb > c->stage_bytes)
	{
		if (c->stage) (void) hipFree(c->stage);
		c->stage = nullptr; c->stage_bytes = 0;
		const size_t want = std::max<size_t>(qb + lb + db + cb, (size_t) 1 << 20);
		HIPCHK(hipMalloc(&c->stage, want));
		c->stage_bytes = want;
	}
	char *p = (char *) c->stage;
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	HIPCHK(hipMemcpyAsync(dq, queries, nq * dim * 4, hipMemcpyHostToDevice, c->stream));
	int rc = launch_search(ix, &c->ws, dq, dim, nq, ef, 0, dl, nullptr, dd, dc, nullptr, c->stream);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost, c->stream));
	if (dists) HIPCHK(hipMemcpyAsync(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(counts, dc, nq * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (size_t i = 0; i < nq; i++)                            // (as hnsw_gpu_search_batch: an interrupted launch is an error of a host-pointer call)
		if (counts[i] == ABORTED_COUNT)
			return fail(HNSW_GPU_ERR_INTERNAL, "the search launch was asked to end early (abort word): query %zu has no result", i);
	return HNSW_GPU_OK;
}

// Streamed completion (hnsw_gpu.h): device-pointer launch on the context's own stream with per-query
// completion flags.  Nothing is copied and nothing is waited for here.
extern "C" int hnsw_gpu_search_batch_ctx_flags(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef,
											   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
											   uint32_t *d_done)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (!d_done) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	HIPCHK(hipSetDevice(c->ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	std::unique_lock<std::recursive_mutex> lock_(c->ix->mu);      // done_next -> launch is one step
	c->ws.done_next = d_done;
	int rc = launch_search(c->ix, &c->ws, d_queries, c->ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats,
						   c->stream);
	c->ws.done_next = nullptr;
	return rc;
}

// ------------------------------------------------------------------------------------
// streams: ONE resident search launch that the host feeds while it runs (device_search.h, "Stream mode")
// ------------------------------------------------------------------------------------
struct hnsw_gpu_stream
{
	hnsw_gpu_ctx *ctx = nullptr;
	size_t ef = 0, ring = 0, dim = 0;
	char *pin = nullptr;                      // pinned, coherent: [queries | labels | dists | counts | flags | control words]
	float *Q = nullptr; label_t *L = nullptr; dist_t *D = nullptr; uint32_t *C = nullptr; uint32_t *F = nullptr;
	uint32_t *host_ctl = nullptr;             // [0] = queries published so far, [1] = stop
	uint32_t *dev_ctl = nullptr;              // the doorbell wave's device copies
	unsigned walkers = 0;
};

extern "C" int hnsw_gpu_stream_close(hnsw_gpu_stream *s);

extern "C" int hnsw_gpu_stream_open(hnsw_gpu_ctx *c, size_t ef, size_t ring, unsigned walkers, hnsw_gpu_stream **out)
{
	if (!c || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ring < 64 || ring > ((size_t) 1 << 20) || (ring & (ring - 1))) return fail(HNSW_GPU_ERR_ARG, "ring must be a power of two in [64, 2^20]");
	if (ef == 0 || ef > 512) return fail(HNSW_GPU_ERR_ARG, "a stream needs ef <= 512 (the team form of the beam kernel)");
	hnsw_gpu_index *ix = c->ix;
	HIPCHK(hipSetDevice(ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	hnsw_gpu_stream *s = new (std::nothrow) hnsw_gpu_stream();
	if (!s) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	s->ctx = c; s->ef = ef; s->ring = ring; s->dim = ix->meta.dim; s->walkers = walkers ? walkers : 4u;
	const size_t qb = round_up(ring * s->dim * 4, 256), lb = round_up(ring * ef * 8, 256), db = round_up(ring * ef * 4, 256),
				 cb = round_up(ring * 4, 256), fb = round_up(ring * 4, 256);
	hipError_t e = hipHostMalloc((void **) &s->pin, qb + lb + db + cb + fb + 256, hipHostMallocCoherent);
	if (e == hipSuccess) e = hipMalloc((void **) &s->dev_ctl, STREAM_COPIES * STREAM_COPY_WORDS * 4);
	if (e == hipSuccess) e = hipMemset(s->dev_ctl, 0, STREAM_COPIES * STREAM_COPY_WORDS * 4);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		if (s->pin) (void) hipHostFree(s->pin);
		if (s->dev_ctl) (void) hipFree(s->dev_ctl);
		delete s;
		return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "stream buffers: %s", hipGetErrorString(e));
	}
	memset(s->pin, 0, qb + lb + db + cb + fb + 256);
	s->Q = (float *) s->pin; s->L = (label_t *) (s->pin + qb); s->D = (dist_t *) (s->pin + qb + lb);
	s->C = (uint32_t *) (s->pin + qb + lb + db); s->F = (uint32_t *) (s->pin + qb + lb + db + cb);
	s->host_ctl = (uint32_t *) (s->pin + qb + lb + db + cb + fb);
	int rc;
	{
		std::unique_lock<std::recursive_mutex> lock_(ix->mu);       // the "next launch only" fields and the launch are one step
		c->ws.done_next = s->F;
		c->ws.stream_host_next = s->host_ctl; c->ws.stream_dev_next = s->dev_ctl;
		c->ws.stream_ring_next = (uint32_t) ring; c->ws.stream_walkers_next = s->walkers;
#ifdef PGEMB_SIMT_EMULATOR
		simt::next_launch_is_resident();                         // (the CPU tier's emulator runs every other launch at the call)
#endif
		rc = launch_search(ix, &c->ws, s->Q, s->dim, ring, ef, 0, s->L, nullptr, s->D, s->C, nullptr, c->stream);
		c->ws.done_next = nullptr; c->ws.stream_host_next = nullptr; c->ws.stream_dev_next = nullptr;
	}
	if (rc)
	{
		(void) hipHostFree(s->pin); (void) hipFree(s->dev_ctl);
		delete s;
		return rc;
	}
	*out = s;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_stream_buffers(hnsw_gpu_stream *s, coord_t **queries, label_t **labels, dist_t **dists, uint32_t **counts, uint32_t **flags)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	if (queries) *queries = s->Q;
	if (labels) *labels = s->L;
	if (dists) *dists = s->D;
	if (counts) *counts = s->C;
	if (flags) *flags = s->F;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_stream_publish(hnsw_gpu_stream *s, uint32_t published_total)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	// Monotonic (a counter mod 2^32, compared by signed difference): several producer threads may publish, and a call that arrives
	// late with a smaller count must not take the word back.  Release: everything written into the slots before is visible before it.
	uint32_t cur = __atomic_load_n(&s->host_ctl[0], __ATOMIC_RELAXED);
	while ((int32_t) (published_total - cur) > 0 &&
		   !__atomic_compare_exchange_n(&s->host_ctl[0], &cur, published_total, true, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) {}
	return HNSW_GPU_OK;
}

// 1 = the stream's launch is still on the device, 0 = it has left (stopped, aborted or failed), < 0 = error
extern "C" int hnsw_gpu_stream_alive(hnsw_gpu_stream *s)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	const int idle = hnsw_gpu_ctx_idle(s->ctx);
	return idle < 0 ? idle : (idle ? 0 : 1);
}

static int stream_end(hnsw_gpu_stream *s, bool keep_buffers);
extern "C" int hnsw_gpu_stream_close(hnsw_gpu_stream *s) { return stream_end(s, false); }
/* the same stop, but the ring stays allocated (leaked on purpose): for a host that could not prove that none of its threads is
 * still reading or writing the ring it was given (hnsw_gpu_stream_buffers) when it had to give the stream up */
extern "C" int hnsw_gpu_stream_abandon(hnsw_gpu_stream *s) { return stream_end(s, true); }

static int stream_end(hnsw_gpu_stream *s, bool keep_buffers)
{
	if (!s) return HNSW_GPU_OK;
	hnsw_gpu_ctx *c = s->ctx;
	(void) hipSetDevice(c->ix->device);
	__atomic_store_n(&s->host_ctl[1], 1u, __ATOMIC_SEQ_CST);
	// every wave leaves at its next look (a walking wave after its query: under a millisecond); a launch that does not is a
	// hung launch: its workspace's abort word, then the wait again
	int rc = HNSW_GPU_OK;
	const int64_t t0 = now_ms();
	bool asked = false;
	while (hipStreamQuery(c->stream) == hipErrorNotReady)
	{
		if (!asked && now_ms() - t0 > 2000)
		{
			std::lock_guard<std::mutex> g(g_ws_mu);
			(void) abort_ws_locked(&c->ws);
			asked = true;
		}
		if (now_ms() - t0 > 1000ll * poll_limit_s()) { rc = fail(HNSW_GPU_ERR_INTERNAL, "the stream's launch did not end"); break; }
		std::this_thread::sleep_for(std::chrono::microseconds(20));
	}
	(void) hipGetLastError();
	if (rc == HNSW_GPU_OK && !keep_buffers)
	{
		(void) hipHostFree(s->pin);
		(void) hipFree(s->dev_ctl);
	}                                                            // (a launch that never ended may still write them: leaked on purpose)
	delete s;
	return rc;
}

// 1 = the context's last launch has left the device, 0 = still running, < 0 = error.
extern "C" int hnsw_gpu_ctx_idle(hnsw_gpu_ctx *c)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (c->ws.launches == 0) return 1;
	const int evi = (int) ((c->ws.launches - 1) % SearchWs::EV_RING);
	hipError_t e = hipEventQuery(c->ws.ev1[evi]);
	if (e == hipSuccess) return 1;
	if (e == hipErrorNotReady) { (void) hipGetLastError(); return 0; }
	return fail(HNSW_GPU_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------
// A device buffer shared between PROCESSES: the exchange buffer of a row-sharded search whose shards live in different
// processes (one GPU-owning server per GPU).  Every process searches its shard with its output pointers inside the buffer
// (hnsw_gpu_search_batch_dev: slot r = the r-th [nq][ef] block), the owner merges (hnsw_gpu_merge_topk_strided_dev) once the
// others have told it — over whatever channel they already share — that their launches are complete.  With the importer on
// another GPU its stores cross xGMI as peer stores, exactly like the one-process form (hnsw_gpu_sharded_search_dev); no
// staging copy, no collective library in a C host.
// ------------------------------------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(hnsw_gpu_ipc_handle), "the ABI's handle must hold a HIP IPC handle");

extern "C" int hnsw_gpu_shared_alloc(int device, size_t bytes, void **d_ptr, hnsw_gpu_ipc_handle *handle)
{
	if (!d_ptr || !handle || bytes == 0) return fail(HNSW_GPU_ERR_ARG, "NULL argument or empty buffer");
	HIPCHK(hipSetDevice(device));
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) { (void) hipGetLastError(); return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
	hipIpcMemHandle_t h;
	e = hipIpcGetMemHandle(&h, p);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		(void) hipFree(p);
		return fail(HNSW_GPU_ERR_HIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set where the driver only has dmabuf IPC?)", hipGetErrorString(e));
	}
	memset(handle, 0, sizeof(*handle));
	memcpy(handle->bytes, &h, sizeof(h));
	*d_ptr = p;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_open(int device, const hnsw_gpu_ipc_handle *handle, void **d_ptr)
{
	if (!d_ptr || !handle) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(device));
	hipIpcMemHandle_t h;
	memcpy(&h, handle->bytes, sizeof(h));
	void *p = nullptr;
	const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
	if (e != hipSuccess) { (void) hipGetLastError(); return fail(HNSW_GPU_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
	*d_ptr = p;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_close(int device, void *d_ptr)
{
	if (!d_ptr) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipIpcCloseMemHandle(d_ptr));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_free(int device, void *d_ptr)
{
	if (!d_ptr) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipFree(d_ptr));
	return HNSW_GPU_OK;
}

// Pinned host memory for the host-pointer entry points (NULL when there is no device / no memory).
extern "C" void *hnsw_gpu_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
	{
		(void) hipGetLastError();
		fail(HNSW_GPU_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

extern "C" void hnsw_gpu_host_free(void *p)
{
	if (p) (void) hipHostFree(p);
}

// ------------------------------------------------------------------------------------
// measured roof of the access pattern (device_roof.h)
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_gather_roof(hnsw_gpu_index *ix, int loads_per_lane, int waves_per_cu, unsigned iters, float *gbps)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !gbps) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ix->n == 0 || iters == 0 || waves_per_cu <= 0 || (waves_per_cu & 3)) return fail(HNSW_GPU_ERR_ARG, "need rows, iters > 0 and waves_per_cu %% 4 == 0");
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t row_f4 = ix->stride / 4;
	const uint32_t blocks = (uint32_t) (ix->num_cu * waves_per_cu / 4);
	float *out = (float *) ix->misc + 8;
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	float best = 1e30f;
	int rc = HNSW_GPU_OK;
	for (int rep = 0; rep < 4 && rc == HNSW_GPU_OK; rep++)
	{
		(void) hipEventRecord(e0, nullptr);
		const float4 *base = (const float4 *) ix->vec;
		switch (loads_per_lane)
		{
#define ROOF(T) case T: hipLaunchKernelGGL(gather_roof_kernel<T>, dim3(blocks), dim3(256), 0, nullptr, base, (uint32_t) ix->n, row_f4, iters, out); break
			ROOF(4); ROOF(8); ROOF(12); ROOF(16); ROOF(24);
#undef ROOF
			default: rc = fail(HNSW_GPU_ERR_ARG, "loads_per_lane must be 4, 8, 12, 16 or 24");
		}
		if (rc) break;
		(void) hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(HNSW_GPU_ERR_HIP, "gather roof kernel failed"); break; }
		float ms = 0.f;
		(void) hipEventElapsedTime(&ms, e0, e1);
		if (rep > 0 && ms < best) best = ms;          // first repetition warms up
	}
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	if (rc) return rc;
	const double bytes = (double) blocks * 4.0 * iters * loads_per_lane * 64.0 * 16.0;
	*gbps = (float) (bytes / best / 1e6);
	return HNSW_GPU_OK;
}

// Replay roof (device_roof.h): the rows a traced launch scored, gathered again by `slots` resident waves in the same query
// order with nothing in between.  d_stats = that launch's stats array ({evals, hops} per query).  *ms = best of 3 timed
// repetitions (after one warm-up), *bytes = row bytes one repetition reads.
extern "C" int hnsw_gpu_replay_roof_parts(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
										  unsigned slots, int kb, int rpg, unsigned parts, float *ms, double *bytes, uint64_t *word_sum);
extern "C" int hnsw_gpu_replay_roof(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
									unsigned slots, int kb, int rpg, float *ms, double *bytes, uint64_t *word_sum)
{
	return hnsw_gpu_replay_roof_parts(ix, d_evals, evals_cap, d_stats, nq, slots, kb, rpg, 1, ms, bytes, word_sum);
}

// The same with every query's trace cut into `parts` equal pieces gathered by different waves: the roof of a launch that gives one
// walk's rows to `parts` waves (fewer queries than resident waves).
extern "C" int hnsw_gpu_replay_roof_parts(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
										  unsigned slots, int kb, int rpg, unsigned parts, float *ms, double *bytes, uint64_t *word_sum)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_evals || !d_stats || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ix->n == 0 || nq == 0 || slots < 4 || evals_cap == 0) return fail(HNSW_GPU_ERR_ARG, "need rows, queries and at least 4 slots");
	if (parts == 0 || parts > 64 || nq * (size_t) parts >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "parts must be 1..64");
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t row_f4 = ix->stride / 4;
	const uint32_t blocks = slots / 4;
	float *out = (float *) ix->misc + 8;
	uint32_t *ticket = ix->misc + 12;
	unsigned long long *d_check = (unsigned long long *) (ix->misc + 14);
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	float best = 1e30f;
	int rc = HNSW_GPU_OK;
	const int shape = kb * 100 + rpg;
	for (int rep = 0; rep < 4 && rc == HNSW_GPU_OK; rep++)
	{
		(void) hipMemsetAsync(ticket, 0, 16, nullptr);           // ticket + the (test-only) word sum behind it
		(void) hipEventRecord(e0, nullptr);
		const float4 *base = (const float4 *) ix->vec;
		switch (shape)
		{
#define ROOF(K, R) case K * 100 + R: \
				if (word_sum) hipLaunchKernelGGL((replay_roof_kernel<K, R, true>), dim3(blocks), dim3(256), 4 * REPLAY_STAGE * 4, nullptr, base, row_f4, d_evals, (uint32_t) evals_cap, d_stats, (uint32_t) nq, (uint32_t) parts, ticket, out, d_check); \
				else hipLaunchKernelGGL((replay_roof_kernel<K, R, false>), dim3(blocks), dim3(256), 4 * REPLAY_STAGE * 4, nullptr, base, row_f4, d_evals, (uint32_t) evals_cap, d_stats, (uint32_t) nq, (uint32_t) parts, ticket, out, d_check); \
				break
			ROOF(2, 2); ROOF(2, 4); ROOF(2, 8); ROOF(4, 2); ROOF(4, 4); ROOF(8, 2); ROOF(12, 1); ROOF(12, 2); ROOF(6, 4);
#undef ROOF
			default: rc = fail(HNSW_GPU_ERR_ARG, "no replay shape <%d, %d> (have <2,2> <2,4> <2,8> <4,2> <4,4> <8,2> <6,4> <12,1> <12,2>)", kb, rpg);
		}
		if (rc) break;
		(void) hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(HNSW_GPU_ERR_HIP, "replay roof kernel failed"); break; }
		float t = 0.f;
		(void) hipEventElapsedTime(&t, e0, e1);
		if (rep > 0 && t < best) best = t;            // first repetition warms up
	}
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	if (rc) return rc;
	*ms = best;
	if (word_sum) HIPCHK(hipMemcpy(word_sum, d_check, 8, hipMemcpyDeviceToHost));   // of the last repetition
	if (bytes)
	{
		// rows actually in the trace: sum over queries of min(evals, cap)
		std::vector<uint32_t> st(2 * nq);
		HIPCHK(hipMemcpy(st.data(), d_stats, 2 * nq * 4, hipMemcpyDeviceToHost));
		double rows = 0;
		for (size_t i = 0; i < nq; i++) rows += (double) std::min<size_t>(st[2 * i], evals_cap);
		*bytes = rows * ix->stride * 4.0;
	}
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// row-sharded index inside ONE process: shards on one or several devices, per-shard searchKnn,
// results written straight into the merge device's memory (peer access over xGMI), one merge kernel
// ------------------------------------------------------------------------------------
struct hnsw_gpu_sharded
{
	std::mutex mu;
	std::vector<hnsw_gpu_index *> shards;
	int home = 0;                                   // device of shard 0: queries arrive and results leave there
	std::vector<hipStream_t> streams;               // one per shard, on the shard's device
	std::vector<SearchWs *> ws;                     // ... and a search workspace of its own per shard: a direct search on a shard
	                                                // (its default workspace) and a sharded call never share tickets or bitmaps
	std::vector<hipEvent_t> done;                   // shard i's results are in the home device's gather buffer (timing enabled)
	hipEvent_t merge_start = nullptr;               // home: every shard's `done` has been waited for, the merge kernel is next
	bool timed = false;                             // a call has completed its enqueue: hnsw_gpu_sharded_last_ms has something to read
	std::vector<bool> direct;                       // the shard's device writes home memory directly
	std::vector<float *> q_local; std::vector<size_t> q_cap;          // query copy on a remote shard's device
	std::vector<char *> out_local; std::vector<size_t> out_cap;       // result block when not `direct`
	hipEvent_t ready = nullptr;
	hipEvent_t merged = nullptr; bool merged_set = false;             // end of the previous call's merge: `gather` may be rewritten after it
	char *gather = nullptr; size_t gather_bytes = 0;                  // home: nshards result blocks
	char *io = nullptr; size_t io_bytes = 0;                          // home: staging of the host-pointer form
	hipStream_t home_stream = nullptr;
};

extern "C" void hnsw_gpu_sharded_destroy(hnsw_gpu_sharded *s)
{
	if (!s) return;
	for (size_t i = 0; i < s->shards.size(); i++)
	{
		(void) hipSetDevice(s->shards[i]->device);
		if (i < s->streams.size() && s->streams[i]) (void) hipStreamDestroy(s->streams[i]);
		if (i < s->ws.size() && s->ws[i]) { ws_free(s->ws[i]); delete s->ws[i]; }
		if (i < s->done.size() && s->done[i]) (void) hipEventDestroy(s->done[i]);
		if (i < s->q_local.size() && s->q_local[i]) (void) hipFree(s->q_local[i]);
		if (i < s->out_local.size() && s->out_local[i]) (void) hipFree(s->out_local[i]);
	}
	(void) hipSetDevice(s->home);
	if (s->ready) (void) hipEventDestroy(s->ready);
	if (s->merged) (void) hipEventDestroy(s->merged);
	if (s->merge_start) (void) hipEventDestroy(s->merge_start);
	if (s->gather) (void) hipFree(s->gather);
	if (s->io) (void) hipFree(s->io);
	if (s->home_stream) (void) hipStreamDestroy(s->home_stream);
	delete s;
}

extern "C" int hnsw_gpu_sharded_create(hnsw_gpu_index *const *shards, size_t nshards, hnsw_gpu_sharded **out)
{
	if (!shards || !out || nshards == 0) return fail(HNSW_GPU_ERR_ARG, "need at least one shard");
	for (size_t i = 0; i < nshards; i++)
	{
		if (!shards[i]) return fail(HNSW_GPU_ERR_ARG, "shard %zu is NULL", i);
		if (shards[i]->meta.dim != shards[0]->meta.dim || shards[i]->meta.dist_func != shards[0]->meta.dist_func)
			return fail(HNSW_GPU_ERR_ARG, "shard %zu differs in dims / metric from shard 0", i);
	}
	hnsw_gpu_sharded *s = new (std::nothrow) hnsw_gpu_sharded();
	if (!s) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	s->shards.assign(shards, shards + nshards);
	s->home = shards[0]->device;
	s->streams.assign(nshards, nullptr); s->done.assign(nshards, nullptr); s->direct.assign(nshards, false);
	s->ws.assign(nshards, nullptr);
	s->q_local.assign(nshards, nullptr); s->q_cap.assign(nshards, 0);
	s->out_local.assign(nshards, nullptr); s->out_cap.assign(nshards, 0);
	hipError_t e = hipSuccess;
	for (size_t i = 0; i < nshards && e == hipSuccess; i++)
	{
		const int dev = shards[i]->device;
		if ((e = hipSetDevice(dev)) != hipSuccess) break;
		if ((e = hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking)) != hipSuccess) break;
		if ((e = hipEventCreate(&s->done[i])) != hipSuccess) break;
		s->ws[i] = new (std::nothrow) SearchWs();
		if (!s->ws[i] || ws_init(s->ws[i]) != HNSW_GPU_OK) { e = hipErrorOutOfMemory; break; }
		if (dev == s->home) s->direct[i] = true;
		else
		{
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, dev, s->home) == hipSuccess && can && (knobs_init(), knob(K_SHARDED_NO_PEER, 0) == 0))
			{
				const hipError_t pe = hipDeviceEnablePeerAccess(s->home, 0);
				if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) s->direct[i] = true;
				(void) hipGetLastError();
			}
		}
	}
	if (e == hipSuccess) e = hipSetDevice(s->home);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreate(&s->merged);
	if (e == hipSuccess) e = hipEventCreate(&s->merge_start);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->home_stream, hipStreamNonBlocking);
	if (e != hipSuccess)
	{
		hnsw_gpu_sharded_destroy(s);
		return fail(HNSW_GPU_ERR_HIP, "sharded index set-up failed: %s", hipGetErrorString(e));
	}
	*out = s;
	return HNSW_GPU_OK;
}

extern "C" size_t hnsw_gpu_sharded_nshards(const hnsw_gpu_sharded *s) { return s ? s->shards.size() : 0; }

static int grow(char **p, size_t *have, size_t want)
{
	if (want <= *have) return HNSW_GPU_OK;
	if (*p) (void) hipFree(*p);
	*p = nullptr; *have = 0;
	HIPCHK(hipMalloc((void **) p, want));
	*have = want;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_sharded_search_dev(hnsw_gpu_sharded *s, const coord_t *d_queries, size_t nq, size_t ef,
										   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, void *stream_)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "sharded index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_queries || !d_labels || !d_counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	std::lock_guard<std::mutex> lk(s->mu);
	hipStream_t stream = (hipStream_t) stream_;
	const size_t ns = s->shards.size(), dim = s->shards[0]->meta.dim;
	// one result block per shard on the home device: [labels nq*ef | dists nq*ef | counts nq]
	const size_t o_d = round_up(nq * ef * 8, 256), o_c = o_d + round_up(nq * ef * 4, 256), block = o_c + round_up(nq * 4, 256);
	HIPCHK(hipSetDevice(s->home));
	int rc = grow(&s->gather, &s->gather_bytes, ns * block);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s->ready, stream));
	for (size_t i = 0; i < ns; i++)
	{
		hnsw_gpu_index *ix = s->shards[i];
		HIPCHK(hipSetDevice(ix->device));
		HIPCHK(hipStreamWaitEvent(s->streams[i], s->ready, 0));
		// a call on ANOTHER user stream than the previous one: its shard kernels must not overwrite `gather` (and the
		// shards' result blocks) while the previous call's merge still reads it
		if (s->merged_set) HIPCHK(hipStreamWaitEvent(s->streams[i], s->merged, 0));
		const float *q = d_queries;
		if (ix->device != s->home)                  // the shard reads its queries from its own HBM
		{
			rc = grow((char **) &s->q_local[i], &s->q_cap[i], nq * dim * 4);
			if (rc) return rc;
			HIPCHK(hipMemcpyPeerAsync(s->q_local[i], ix->device, d_queries, s->home, nq * dim * 4, s->streams[i]));
			q = s->q_local[i];
		}
		char *blk = s->gather + i * block;
		if (!s->direct[i])
		{
			rc = grow(&s->out_local[i], &s->out_cap[i], block);
			if (rc) return rc;
			blk = s->out_local[i];
		}
		// per-shard searchKnn (hnswalg.cpp:234-252); with peer access the kernel's result stores land in the
		// home device's memory directly — no copy step, no collective
		rc = launch_search(ix, s->ws[i], q, dim, nq, ef, 0, (uint64_t *) blk, nullptr, (float *) (blk + o_d), (uint32_t *) (blk + o_c),
						   nullptr, s->streams[i]);
		if (rc) return rc;
		if (!s->direct[i])
			HIPCHK(hipMemcpyPeerAsync(s->gather + i * block, s->home, blk, ix->device, block, s->streams[i]));
		HIPCHK(hipEventRecord(s->done[i], s->streams[i]));
	}
	HIPCHK(hipSetDevice(s->home));
	for (size_t i = 0; i < ns; i++) HIPCHK(hipStreamWaitEvent(stream, s->done[i], 0));
	HIPCHK(hipEventRecord(s->merge_start, stream));
	rc = hnsw_gpu_merge_topk_strided_dev(s->home, (const label_t *) s->gather, block / 8, (const dist_t *) (s->gather + o_d), block / 4,
										 ns, nq, ef, d_labels, d_dists, d_counts, stream);
	if (rc) return rc;
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipEventRecord(s->merged, stream));
	s->merged_set = true;
	s->timed = true;
	return HNSW_GPU_OK;
}

// Where the time of the last hnsw_gpu_sharded_search[_dev] call went, per shard, from HIP events on each shard's own device:
// search_ms[i] = shard i's search kernel, peer_ms[i] = what followed it on that shard's stream until its results were in the home
// device's buffer (0 when the kernel stores them there itself through peer access: then the xGMI stores are part of search_ms;
// otherwise the staged peer copy), *merge_ms = the merge kernel on the home device.  Arrays of hnsw_gpu_sharded_nshards values;
// waits for the call to finish.
extern "C" int hnsw_gpu_sharded_last_ms(hnsw_gpu_sharded *s, float *search_ms, float *peer_ms, float *merge_ms, int *direct)
{
	if (!s || !search_ms || !peer_ms || !merge_ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(s->mu);
	if (!s->timed) return fail(HNSW_GPU_ERR_ARG, "no sharded search has run yet");
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipEventSynchronize(s->merged));
	HIPCHK(hipEventElapsedTime(merge_ms, s->merge_start, s->merged));
	for (size_t i = 0; i < s->shards.size(); i++)
	{
		SearchWs *w = s->ws[i];
		if (w->launches == 0) return fail(HNSW_GPU_ERR_INTERNAL, "shard %zu has no launch on record", i);
		const int evi = (int) ((w->launches - 1) % SearchWs::EV_RING);
		HIPCHK(hipSetDevice(s->shards[i]->device));
		HIPCHK(hipEventSynchronize(s->done[i]));
		HIPCHK(hipEventElapsedTime(&search_ms[i], w->ev0[evi], w->ev1[evi]));
		HIPCHK(hipEventElapsedTime(&peer_ms[i], w->ev1[evi], s->done[i]));
		if (direct) direct[i] = s->direct[i] ? 1 : 0;
	}
	HIPCHK(hipSetDevice(s->home));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_sharded_search(hnsw_gpu_sharded *s, const coord_t *queries, size_t nq, size_t ef,
									   label_t *labels, dist_t *dists, uint32_t *counts)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "sharded index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	const size_t dim = s->shards[0]->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	char *p;
	{
		std::lock_guard<std::mutex> lk(s->mu);
		HIPCHK(hipSetDevice(s->home));
		int rc = grow(&s->io, &s->io_bytes, qb + lb + db + cb);
		if (rc) return rc;
		p = s->io;
	}
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	HIPCHK(hipMemcpyAsync(dq, queries, nq * dim * 4, hipMemcpyHostToDevice, s->home_stream));
	int rc = hnsw_gpu_sharded_search_dev(s, dq, nq, ef, dl, dd, dc, s->home_stream);
	if (rc) return rc;
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipMemcpyAsync(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost, s->home_stream));
	if (dists) HIPCHK(hipMemcpyAsync(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost, s->home_stream));
	HIPCHK(hipMemcpyAsync(counts, dc, nq * 4, hipMemcpyDeviceToHost, s->home_stream));
	HIPCHK(hipStreamSynchronize(s->home_stream));
	return HNSW_GPU_OK;
}
