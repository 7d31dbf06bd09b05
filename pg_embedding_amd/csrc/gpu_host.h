// gpu_host.h — what the translation units of libhnsw_gpu.so share on the HOST side: error text, the knob table, the search
// workspace, the mirror object and the few internal entry points one unit offers the others.  Nothing in here is part of the
// C ABI (include/hnsw_gpu.h, include/hnsw_gpu_diag.h); everything has hidden visibility.
//
//   hnsw_gpu.hip      errors, configuration, workspaces + watchdog, the mirror (create / import / export / append / reserve)
//   gpu_search.hip    launch planning (launch_search), the search entry points, traces of one walk, search contexts
//   gpu_stream.hip    streams: one resident launch fed by the host
//   gpu_scan.hip      batched distances, exhaustive k-NN (canonical scan, MFMA filter)
//   gpu_build.hip     insert path: batched link step, single inserts
//   gpu_sharded.hip   top-k merge, shards in one process, the exchange buffer shared between processes
//   gpu_diag.hip      measurement only (include/hnsw_gpu_diag.h): traced launches, replay / gather roofs, clocks, placement
//   search_inst.hip   the search kernels, one load shape per unit;  sort_pairs.hip  the batched build's pair sort
#pragma once
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
#include "hnsw_gpu_diag.h"
#include "search_kernels.h"

using namespace pgemb;

#pragma GCC visibility push(hidden)

// ---- errors ----------------------------------------------------------------------------
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);

#define HIPCHK(expr)                                                                          \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return fail(HNSW_GPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
						__FILE__, __LINE__);                                                  \
	} while (0)

static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ---- configuration: resolved ONCE per process, never on a call path (hnsw_gpu.hip) ---------
enum Knob : int
{
	// operational (environment, read once)
	K_BEAM, K_FORCE_LDS_HEAPS, K_TEAM, K_TEAM_MAX_NQ, K_WIDE_EF_MIN, K_REF_ORDER, K_NO_POLL, K_POLL_LIMIT_S, K_INSERT_FUSED,
	K_BLOCKS_PER_CU, K_STREAM_LIGHT,
	// test knobs (hnsw_gpu_config_set only)
	K_BEAM16, K_NARROW5, K_LEAN, K_HASH_ENTRIES, K_LDS_SET_MIN_WAVES, K_TEAM_SPEC, K_TEAM_WPB, K_NARROW_WPB, K_ABORT_POLL_LOG2, K_MAX_BLOCKS, K_SHARDED_NO_PEER, K_BF_BIG_MIN_BLOCKS,
#ifdef HNSW_EXPERIMENT
	K_WIDE_WAVES, K_SHAPE_12X1, K_TEAM_MAINS, K_TEAM_COUNTERS,
#endif
	K_COUNT
};
struct KnobVal { std::atomic<long long> v{0}; std::atomic<bool> set{false}; };
extern KnobVal g_knob[K_COUNT];
void knobs_init();

// value of knob k, or `dflt` when nobody set it
static inline long long knob(int k, long long dflt)
{
	return g_knob[k].set.load(std::memory_order_acquire) ? g_knob[k].v.load(std::memory_order_relaxed) : dflt;
}
static inline bool knob_is_set(int k) { return g_knob[k].set.load(std::memory_order_acquire); }

// ---- the search workspace -------------------------------------------------------------------
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

extern std::mutex &g_ws_mu;                              // guards the registry of workspaces (abort + watchdog, hnsw_gpu.hip)
int64_t now_ms();
int abort_ws_locked(SearchWs *w);                        // g_ws_mu held
int ws_init(SearchWs *w);
void ws_free(SearchWs *w);

// ---- the device mirror ------------------------------------------------------------------------
struct hnsw_gpu_index
{
	// One search / build / scratch user at a time per mirror: the public entry points that touch
	// the shared workspace take this lock (launches stay asynchronous on the caller's stream, but
	// two host threads must not interleave their launches on one handle).
	std::recursive_mutex mu;
	HnswMetadata meta;
	int      device = 0;
	int      num_cu = 0;
	bool     gfx950 = false;        // the device's gcnArchName says so (the MFMA filter's direct-to-LDS loads and LDS sizes are gfx950's)
	size_t   max_lds = 64 * 1024;   // dynamic LDS one block may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
	bool     ins_dirty = false;     // an insert failed after its kernels were enqueued: block counters may be non-zero (insert_impl)
	size_t   n = 0, cap = 0;
	uint32_t stride = 0;      // floats per row (dim rounded up to 4)
	uint32_t lstride = 0;     // link slots per element (maxM rounded up to 16)
	// vec | links | labels live in ONE allocation (`arena`), each array starting on a 2 MiB boundary: where the mirror lands in the
	// device's address space does not depend on what the process allocated before (alloc_mirror, hnsw_gpu.hip)
	char     *arena = nullptr; size_t arena_bytes = 0;
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

int ensure_scratch(hnsw_gpu_index *ix, size_t bytes);
int import_range(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count, size_t n_total);

// ---- search (gpu_search.hip) ----------------------------------------------------------------------
static const size_t LDS_PER_CU = 160 * 1024;
int launch_search(hnsw_gpu_index *ix, SearchWs *w, const float *d_queries, size_t q_stride, size_t nq, size_t ef, int mode,
				  uint64_t *d_labels, uint32_t *d_idx, float *d_dists, uint32_t *d_counts, uint32_t *d_stats, hipStream_t stream);
int poll_limit_s();
int poll_done_flag(const volatile uint32_t *flag, const char *what, SearchWs *w);
int ws_search_ms(int device, SearchWs *w, unsigned back, float *ms);

// search contexts: independent batches in flight on different streams
struct hnsw_gpu_ctx
{
	hnsw_gpu_index *ix;
	SearchWs ws;
	// host-pointer form (hnsw_gpu_search_batch_ctx_host): own stream + device staging, grow-only
	hipStream_t stream = nullptr;
	void *stage = nullptr; size_t stage_bytes = 0;
};

#pragma GCC visibility pop
