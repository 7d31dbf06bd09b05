// device_search.h — fused traverse + score kernels: searchBaseLayer (hnswalg.cpp:42-114),
// searchKnn (hnswalg.cpp:234-252) and the result ordering of hnsw_search
// (hnswalg.cpp:256-277) for a batch of independent queries.
//
// Mapping.  ONE WAVEFRONT = ONE QUERY AT A TIME.  A launch is a fixed set of resident
// waves ("slots"); each slot pulls query numbers from an atomic ticket until the batch
// is exhausted, so long and short traversals balance.  Throughput comes from thousands of
// slots each keeping up to 24 KiB of row reads in flight; a single query is latency-bound by
// construction (≈ef dependent hops, two memory round trips each).
//
// Three forms of the same algorithm (the host picks one per launch, hnsw_gpu.hip launch_search):
//   hnsw_search_kernel_beam (ef <= 256; <= 512 on wide rows; the hot one): ONE unordered set of accepted
//                           elements in registers, every decision of the reference restated as a count
//                           over it (banner further down); visited set = exact bucketed tag set in LDS
//                           (banner at tagset_split) with the HBM bitmap behind it;
//   hnsw_search_kernel_reg  (ef <= 256, fallback): result set sorted + candidate set unsorted, both in
//                           registers;
//   hnsw_search_kernel_lds  (any ef): both sets as unsorted arrays, in LDS or (large ef) in HBM;
//                           visited set = bitmap.
//
// Heaps.  The reference keeps two std::priority_queue<pair<float,idx>>:
//   topResults  : max-heap on ( dist, idx)   -> worst on top, evicted when size > ef
//   candidateSet: max-heap on (-dist, idx)   -> best on top, popped to expand
// Only the extremes of those strict total orders are observable, so any container that
// yields the same extremes gives the same traversal.  Keys are 64-bit:
//   results   : ord(dist)<<32 |  idx   (largest key  = topResults.top())
//   candidates: ord(dist)<<32 | ~idx   (smallest key = candidateSet.top(): nearest, ties by
//                                       LARGER idx, as the (-dist, idx) pair order dictates)
// (ord() maps float order onto unsigned order).  The candidate set is BOUNDED without
// changing the result: a candidate whose distance exceeds the current bound can never be
// expanded (the bound only shrinks once results are full, hnswalg.cpp:70,107), and live
// candidates number < 2*ef (<= ef still in results + <= ef-1 evicted at exactly the bound),
// so with capacity >= 2*ef the largest key of an overfull set is always dead and may be dropped.
//
// Visited set (hnswalg.cpp:45-50,82-93: a growable bitmap in the reference).
//   * beam form: bucketed tag set in LDS — 16-byte buckets of eight 16-bit tags, bucket and tag together are
//     the id; one ds_read_b128 tests, one ds_cmpst inserts, whatever the fill; an id whose bucket is full
//     lives in the HBM bitmap instead (exact, see the banner at tagset_split);
//   * two-set register form: LDS hash set of 32-bit ids, open addressing, lock-free ds_cmpst insert = the
//     test and the set of :91-93 in one LDS operation;
//   * per-slot bitmap in HBM: returning atomic OR (safe when two neighbours share a word), bits
//     undone through a log after the query.  It is the only set of the generic form; in the two-set
//     register form it takes over from the hash set once that is 3/4 full (wide rows: both are consulted
//     from then on; narrow rows: the hash set is flushed into the bitmap once).
// Link lists are de-duplicated at upload (first occurrence kept), which is behaviour-preserving
// because a repeated id is always already visited when reached again in pass 2 (:89-93).
#pragma once
#include <type_traits>
#include "device_dist.h"

namespace pgemb {

constexpr uint32_t LINK_NONE = 0xFFFFFFFFu;

struct SearchArgs
{
	// index mirror
	const float    *vec;        // rows, `stride` floats apart, zero padded
	const uint32_t *links;      // lstride ids per element, LINK_NONE padded
	const uint64_t *labels;
	uint32_t n, dim, stride, nchunks, kiters, maxM, lstride, entry;
	// batch
	const float *queries;       // query i at queries + i * q_stride (dim floats used)
	uint32_t q_stride;
	uint32_t nq, ef, ccap;      // ccap = 2*ef candidate capacity (LDS form)
	// outputs
	uint64_t *out_labels;       // mode 0: nq*ef
	uint32_t *out_idx;          // mode 1: nq*ef
	float    *out_dists;        // nq*ef or null
	uint32_t *out_counts;       // nq
	uint32_t *out_stats;        // nq*2 {evals, hops} or null
	uint32_t *out_pops;         // null, or nq * pops_cap element numbers: the walk's pop sequence (hnswalg.cpp:73), for callers that
	uint32_t  pops_cap;         // validate a walk against the host (hnsw_gpu_search_trace); out_stats' hop count says how many
	uint32_t *out_evals;        // null, or nq * evals_cap element numbers: the rows the walk scored, in scoring order (out_stats' evaluation
	uint32_t  evals_cap;        // count says how many), and out_times[2*qi], [2*qi+1] = s_memrealtime at the query's start and at the end of
	uint64_t *out_times;        // its walk — the trace the replay roof and the reuse analysis of bench.py are made from (hnsw_gpu_search_traced_dev)
	uint32_t *done;             // null, or nq completion flags (host-visible): 1 is stored with system scope
	                            // once query i's outputs are complete (hnsw_gpu_search_batch_ctx_flags)
	// per-slot workspace
	uint32_t *vis;              // slots * vis_words, all zero between queries
	uint32_t *vlog;             // slots * logcap
	uint64_t  vis_words;
	uint32_t  logcap;
	uint32_t *ticket;           // zeroed before every launch
	// LDS carve (bytes, per wave)
	uint32_t qpad_floats, off_res, off_cand, off_newid, off_newdist, wave_bytes;
	// register form only: exact visited hash set in LDS (power-of-two entries, 0 = off); ids that
	// arrive after it is hmax full go to the HBM bitmap instead
	uint32_t off_hash, hcap, hmax;
	uint32_t hmagic;            // beam form: ceil(2^38 / buckets) of the bucketed set (hcap / 4 buckets of eight 16-bit tags)
	uint64_t *beam_scratch;     // (unused since round 6: the beam form's prune compacts in registers; rounds 1-5: a per-slot HBM scratch line)
	uint64_t *set_scratch;      // generic form with its sets in HBM: per-slot area, set_stride keys apart
	uint32_t out_stride;        // result slots per query in the output arrays (the caller's ef; a.ef may be clamped to n)
	size_t set_stride;
	uint32_t wide_p, wide_ch, wide_nr, wide_nc;   // wide-beam form (device_search_wide.h): result slots (a power of two >= ef), keys per
	                            // chunk, chunks of the result / candidate array (their extremes live in LDS at off_res / off_cand)
	int mode;                   // 0 = hnsw_search semantics, 1 = searchBaseLayer only
	// team form (beam kernel, TEAM = true; banner "Team form" further down)
	uint32_t team_mains;        // waves of a block that take queries (wib < team_mains); the others start as helpers
	uint32_t off_ctl;           // byte offset of the block's TeamCtl array in dynamic LDS (behind the wave regions)
	uint32_t tm_off_ex, tm_off_miss, tm_off_lctag, tm_off_lcstate, tm_off_lclinks, tm_lcslots, tm_off_dc, tm_dccap;
	uint32_t tm_spec;           // helpers of rank < tm_spec speculate (packages); the others score slices of the walking wave's rows
	const uint32_t *abort_word; // null, or the workspace's abort word in pinned host memory (banner at abort_requested)
	uint32_t abort_mask;        // a wave looks at the abort word at the top of query number qi when (qi & abort_mask) == 0: every 16th by default
	uint32_t *health;           // HEALTH_WORDS device words: time-out and abort counters of the workspace
	uint32_t *team_dbg;         // null, or 16 counters for the whole launch (hnsw_gpu_team_counters): hops with helpers,
	                            // link-list hits, ids looked up, distance hits, hops that still scored rows, all hops
	// stream mode (team form of the beam kernel; banner "Stream mode" at the kernel): a RESIDENT launch that the host feeds
	const uint32_t *stream_host;  // null = a plain launch.  Pinned host words: [0] = queries published so far, [1] = non-zero: leave
	uint32_t *stream_dev;         // device: STREAM_COPIES copies of the two words, 128 bytes apart (the doorbell wave's 64 lanes store one each):
	                              // wave w polls copy w % STREAM_COPIES, so that the idle waves of a stream do not all read ONE line of ONE L2 channel
	uint32_t stream_ring;         // slots of the query / result ring (a power of two): ticket t lives in slot t & (ring - 1)
	uint32_t stream_light;        // 1 = a stream's results are written with system-scope stores and its completion flag follows vmcnt(0);
	                              // 0 = plain stores + a full system-scope release per answered query (signal_done)
};

// Abort word + health words of a search workspace.
//   abort_word (PINNED HOST memory, so the host sets it with a plain store from any thread and the device's read can never be
//     served from a cache): non-zero = the host asks the launches of this workspace to end (hnsw_gpu_index_abort / the
//     library's watchdog).  Every wave looks at the top of a query and every 256 hops of a walk — a read across the host
//     link, ~2 us, a fraction of a percent of a walk — and leaves.  Outputs of that launch are undefined and the workspace
//     is re-zeroed before the next one.  Nothing in the kernels waits without a bound, so this is for the unknown: a hung
//     launch costs its caller's timeout, not the device.
//     Round 6: NOT at the top of every query any more, but of every 16th (abort_mask, by query number: no state in the wave).  Each look
//     is an uncached read across the host link, and 40 000 narrow-row queries per 5.8 ms launch are 7 million such reads a second from
//     5 120 waves: how fast the host serves them depends on where that pinned page sits, and with the wrong page EVERY launch of a
//     workspace was 40 % slower — the "slow state" of rounds 5-6 (profiles/r6k_*: same process, same index, same kernel, same stream; one
//     workspace 5.8 ms, another 8.6 ms).  The wide-row launches ask a quarter as often per second and never showed it.  An abort now takes
//     effect within 16 queries of every wave (~10-20 ms) instead of one.
//   health (device memory, totals of the workspace's life, all zero in a healthy one):
//   [HEALTH_SLICE_TIMEOUTS]   team form: slices a helper did not deliver within SLICE_WAIT_POLLS (scored by the walking wave)
//   [HEALTH_PACKAGE_TIMEOUTS] team form: packages still "claimed" after 4000 polls (fetched by the walking wave)
//   [HEALTH_ABORTED_WAVES]    waves that left a launch because of the abort word
//   [HEALTH_SLICES_DELIVERED] team form: slices helpers scored for walking waves (says the mechanism is in use; one
//                             non-returning atomic per job)
//   [HEALTH_CLOCK .. +7]      not a health word: four 64-bit clock readings of the LAST launch's first wave (beam kernels) — shader clock and
//                             constant 100 MHz clock when it started, the same two when it left — from which the host computes the shader
//                             clock the launch ran at (hnsw_gpu_last_search_clock_mhz, include/hnsw_gpu_diag.h).  The narrow-row kernel is
//                             bound by instruction issue: its time scales with that clock, and a process that finds the device at a lower
//                             clock sees every launch slower by the same factor.  Two scalar loads + one store pair per LAUNCH.
constexpr uint32_t STREAM_COPIES = 64, STREAM_COPY_WORDS = 32;      // stream mode: copies of the control words, words between two copies
constexpr uint32_t ABORTED_COUNT = 0xFFFFFFFFu;     // out_counts[i] of a query an aborted launch did not answer (include/hnsw_gpu.h)
enum : uint32_t { HEALTH_SLICE_TIMEOUTS = 1, HEALTH_PACKAGE_TIMEOUTS = 2, HEALTH_ABORTED_WAVES = 3, HEALTH_SLICES_DELIVERED = 4, HEALTH_CLOCK = 8, HEALTH_WORDS = 16 };

// the launch's first wave stamps both clocks into the health words: `which` = 0 at its start, 1 when it leaves
__device__ __forceinline__ void clock_stamp(uint32_t *health, uint32_t slot, int lane, int which)
{
	if (slot != 0 || lane != 0 || !health) return;
	uint64_t *c = reinterpret_cast<uint64_t *>(health + HEALTH_CLOCK) + 2 * which;
	c[0] = __builtin_amdgcn_s_memtime();
	c[1] = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ bool abort_word_set(const uint32_t *abort_word)
{
	if (!abort_word) return false;
	return __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u;
}
__device__ __forceinline__ bool abort_requested(const SearchArgs &a) { return abort_word_set(a.abort_word); }

// Streamed completion: everything this wave wrote for the query becomes visible system-wide, then
// the flag.  Once per query, outside the hop loop.
// written_through = every result of this query was written with SYSTEM-SCOPE stores (write-through: nothing of them is held in the
// device's L2) into the library's own coherent pinned ring (a stream): then "every store of this wave has been acknowledged" (vmcnt 0)
// orders them before the flag, and the L2 write-back of a system-scope release — buffer_wbl2, per ANSWERED QUERY, of everybody's
// lines — is not needed.  That write-back was the stream server's ceiling: 0.67 -> 0.75 M q/s at 1 024 backends, 0.49 -> 0.61 M at
// 2 048.  With PLAIN result stores the shortcut is unsound and was measured to be: 15 % of the answers the host read were not the
// query's (profiles/r4q_stream_completion_forms.txt).  Caller-provided buffers keep plain stores + the full release.
__device__ __forceinline__ void signal_done(uint32_t *flag, int lane, bool written_through = false)
{
	if (written_through)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0);
	}
	else
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope
	if (lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t ord_f32(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o)
{
	return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// (used by the exhaustive scorers' top-k lists)
// Insert `key` into ascending A[0..sz); A has cap+1 slots.  If the array was full the
// largest key falls off.  Wave-cooperative, wave-uniform arguments.  Returns new size.
__device__ __forceinline__ uint32_t sorted_insert(uint64_t *A, uint32_t sz, uint64_t key, uint32_t cap, int lane)
{
	uint32_t p = 0;
	for (uint32_t b = 0; b < sz; b += 64)
	{
		uint32_t i = b + lane;
		bool lt = (i < sz) && (A[i] < key);
		p += (uint32_t) __builtin_popcountll(__ballot(lt));
	}
	if (p < sz)
	{
		// move A[p..sz) up by one, highest 64-chunk first so nothing is overwritten early
		for (int b = (int) (sz & ~63u); b >= (int) (p & ~63u); b -= 64)
		{
			uint32_t i = (uint32_t) b + lane;
			bool mv = (i > p) && (i <= sz);
			uint64_t tmp = 0;
			if (mv) tmp = A[i - 1];
			wave_sync();
			if (mv) A[i] = tmp;
			wave_sync();
		}
	}
	if (lane == 0) A[p] = key;
	wave_sync();
	sz += 1;
	return sz > cap ? cap : sz;
}



// =====================================================================================
// Register-resident form (ef <= 64*RREG, RREG in {2,4}): the hot configuration.
//
//   results    : SORTED ascending in RREG 64-bit registers per lane (index = reg*64 + lane,
//                unused slots = ~0).  Insert = ballot-count for the position + one DPP
//                wave_shr:1 shift; the worst element falls off the end.  No LDS, no waits.
//   candidates : UNSORTED in 2*RREG registers per lane (capacity 128*RREG >= 2*ef, which the
//                header comment proves sufficient).  Append = one lane write; pop-best = a
//                per-lane min + a DPP wave-min of the distance word (ties on the distance
//                resolved by a second min over ~idx).
//   accept loop: lane r holds (dist, id) of new row r; rows that cannot beat the bound as
//                it stood at the start of the hop are masked out with one ballot (the bound
//                only decreases, so they would be rejected at their turn anyway) and the
//                survivors are visited in link order via v_readlane.
// =====================================================================================

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v)
{
	return (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) v, CTRL, ROW_MASK, 0xF, false);
}

// min over the wavefront, returned uniformly
// (`old` = the identity of min: lanes a masked row leaves unwritten see the identity, and hipcc folds each step into ONE
// v_min_u32_dpp instead of v_mov + v_mov_dpp + v_min)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
	constexpr uint32_t ID = 0xFFFFFFFFu;
	v = min(v, dpp_u32<0xB1>(ID, v));           // lane ^ 1
	v = min(v, dpp_u32<0x4E>(ID, v));           // lane ^ 2
	v = min(v, dpp_u32<0x141>(ID, v));          // row_half_mirror
	v = min(v, dpp_u32<0x140>(ID, v));          // row_mirror: every lane of a row = row min
	v = min(v, dpp_u32<0x142, 0xA>(ID, v));     // row_bcast15 into rows 1 and 3
	v = min(v, dpp_u32<0x143, 0xC>(ID, v));     // row_bcast31 into rows 2 and 3
	return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, (int) l);
	const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), (int) l);
	return ((uint64_t) hi << 32) | lo;
}

#ifdef HNSW_EXPERIMENT      // (helpers of the two-set register form: experiment builds only, see hnsw_search_kernel_reg)
// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v, uint64_t fill)
{
	const uint32_t lo = dpp_u32<0x138>((uint32_t) fill, (uint32_t) v);              // wave_shr:1
	const uint32_t hi = dpp_u32<0x138>((uint32_t) (fill >> 32), (uint32_t) (v >> 32));
	return ((uint64_t) hi << 32) | lo;
}

// Insert into the sorted result registers.  Only registers at or above the insert position
// change.  Capacity is 64*R >= ef: elements pushed past index ef-1 are not cleared — they were the
// maximum when they fell off and the valid maximum only decreases afterwards, so they stay above
// every valid key, never count in `p` for an accepted key, and are never read (rsize bounds all reads).
template <int R>
__device__ __forceinline__ void res_insert(uint64_t (&rk)[R], uint64_t key, int lane)
{
	uint32_t p = 0;
#pragma unroll
	for (int k = 0; k < R; k++) p += (uint32_t) __builtin_popcountll(__ballot(rk[k] < key));
#pragma unroll
	for (int k = R - 1; k >= 0; k--)
	{
		if (p < (uint32_t) (k + 1) * 64)          // wave-uniform: registers below the position are untouched
		{
			const uint64_t fill = (k > 0) ? readlane_u64(rk[k > 0 ? k - 1 : 0], 63) : 0ull;
			const uint64_t prev = wave_shr1_u64(rk[k], fill);
			const uint32_t i = (uint32_t) k * 64 + lane;
			rk[k] = (i < p) ? rk[k] : ((i == p) ? key : prev);
		}
	}
}

// NOTE: every access below touches ALL registers of the array with compile-time indices and
// picks by select.  Writing `if (k == sel) a[k] = ...` lets the optimiser fold the unrolled
// chain back into a dynamically indexed a[sel], which forces the array into scratch memory.
template <int R>
__device__ __forceinline__ uint64_t res_at(const uint64_t (&rk)[R], uint32_t i)
{
	uint64_t out = 0;
#pragma unroll
	for (int k = 0; k < R; k++)
	{
		const uint64_t t = readlane_u64(rk[k], i & 63);
		out = ((i >> 6) == (uint32_t) k) ? t : out;
	}
	return out;
}

template <int C>
__device__ __forceinline__ void cand_set(uint64_t (&ck)[C], uint32_t slot, uint64_t key, int lane)
{
#pragma unroll
	for (int k = 0; k < C; k++)
	{
		const bool hit = slot == ((uint32_t) k * 64 + (uint32_t) lane);
		ck[k] = hit ? key : ck[k];
	}
}

// Smallest key of the set; returns its slot through `slot`.  Set must be non-empty.
template <int C>
__device__ __forceinline__ uint64_t cand_min(const uint64_t (&ck)[C], uint32_t &slot)
{
	uint64_t m = ck[0];
	uint32_t mk = 0;
#pragma unroll
	for (int k = 1; k < C; k++)
	{
		const bool lt = ck[k] < m;
		m = lt ? ck[k] : m;
		mk = lt ? (uint32_t) k : mk;
	}
	const uint32_t h = (uint32_t) (m >> 32);
	const uint32_t hmin = wave_min_u32(h);
	uint64_t eq = __ballot(h == hmin);
	if (__builtin_popcountll(eq) > 1)                      // equal distances: larger idx first
	{
		const uint32_t lo = (h == hmin) ? (uint32_t) m : 0xFFFFFFFFu;
		const uint32_t lomin = wave_min_u32(lo);
		eq = __ballot(h == hmin && lo == lomin);
	}
	const uint32_t L = (uint32_t) __builtin_ctzll(eq);
	slot = ((uint32_t) __builtin_amdgcn_readlane((int) mk, (int) L) << 6) | L;
	return readlane_u64(m, L);
}

// Largest real key (set full): used only to make room when the candidate set overflows.
template <int C>
__device__ __forceinline__ uint64_t cand_max(const uint64_t (&ck)[C], uint32_t &slot)
{
	uint64_t m = ck[0];
	uint32_t mk = 0;
#pragma unroll
	for (int k = 1; k < C; k++)
	{
		const bool gt = ck[k] > m;
		m = gt ? ck[k] : m;
		mk = gt ? (uint32_t) k : mk;
	}
	const uint32_t h = ~(uint32_t) (m >> 32);
	const uint32_t hmin = wave_min_u32(h);
	uint64_t eq = __ballot(h == hmin);
	if (__builtin_popcountll(eq) > 1)
	{
		const uint32_t lo = (h == hmin) ? ~(uint32_t) m : 0xFFFFFFFFu;
		const uint32_t lomin = wave_min_u32(lo);
		eq = __ballot(h == hmin && lo == lomin);
	}
	const uint32_t L = (uint32_t) __builtin_ctzll(eq);
	slot = ((uint32_t) __builtin_amdgcn_readlane((int) mk, (int) L) << 6) | L;
	return readlane_u64(m, L);
}


#endif

// ---- exact visited set in LDS: open addressing, linear probing, lock-free insert ------------
constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;     // never a valid element number (LINK_NONE)

__device__ __forceinline__ uint32_t hash_slot(uint32_t id, uint32_t mask)
{
	return ((id * 2654435761u) >> 7) & mask;
}

#ifdef HNSW_EXPERIMENT
// true if `id` was not in the table (and is now); per-lane, lanes may collide on a slot
__device__ __forceinline__ bool hash_test_and_set(uint32_t *tab, uint32_t mask, uint32_t id)
{
	uint32_t s = hash_slot(id, mask);
	for (;;)
	{
		const uint32_t old = atomicCAS(&tab[s], HASH_EMPTY, id);
		if (old == HASH_EMPTY) return true;
		if (old == id) return false;
		s = (s + 1) & mask;
	}
}

__device__ __forceinline__ bool hash_contains(const uint32_t *tab, uint32_t mask, uint32_t id)
{
	uint32_t s = hash_slot(id, mask);
	for (;;)
	{
		const uint32_t v = tab[s];
		if (v == id) return true;
		if (v == HASH_EMPTY) return false;
		s = (s + 1) & mask;
	}
}

#endif

// ---- exact visited set in LDS, bucketed (beam form): 16-byte buckets of eight 16-bit tags ---------------
// bucket = id % nb, tag = id / nb + 1 (0 = free slot): bucket and tag together ARE the id, so the set is
// exact.  A bucket fills front to back and never shrinks within a query, so ONE ds_read_b128 answers "seen?" and one
// ds_cmpst on the word that holds the first free slot inserts — two LDS round trips whatever the fill, where linear
// probing of 32-bit ids cost the slowest lane's chain (measured 2.2-2.6 k cycles per hop, profiles/r2k_*).  Twice the
// entries in the same LDS; an id whose bucket is full goes to the HBM bitmap (exact: an id is only ever looked up in
// the bitmap when its bucket is full, and was only ever put there when its bucket was full, and buckets never shrink).
enum : int { TS_SEEN = 0, TS_NEW = 1, TS_FULL = 2, TS_AGAIN = 3 };

// nonzero iff one of the eight 16-bit halves of w equals the tag replicated in tt (the classic zero-field test on
// w ^ tt, exact for "is there one"); straight-line on purpose: `||` of four tests compiles to four divergent branches
__device__ __forceinline__ uint32_t tag_match(const uint4 &w, uint32_t tt)
{
	const uint32_t x0 = w.x ^ tt, x1 = w.y ^ tt, x2 = w.z ^ tt, x3 = w.w ^ tt;
	return (((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) | ((x3 - 0x00010001u) & ~x3)) & 0x80008000u;
}

// id -> (bucket, tag) for any bucket count nb in [128, 1024]: q = id / nb through a 38-bit reciprocal (magic =
// ceil(2^38 / nb); exact for id < 2^28, and the host only enables the set while (n - 1) / nb + 1 fits 16 bits)
__device__ __forceinline__ void tagset_split(uint32_t id, uint32_t nb, uint32_t magic, uint32_t &b, uint32_t &tag)
{
	const uint32_t q = (uint32_t) (((uint64_t) id * magic) >> 38);
	b = id - q * nb;
	tag = q + 1u;
}

__device__ __forceinline__ int tagset_test_and_set(uint32_t *tab, uint32_t nb, uint32_t magic, uint32_t id)
{
	uint32_t b, tag;
	tagset_split(id, nb, magic, b, tag);
	const uint32_t tt = tag * 0x00010001u;
	int st;
	do
	{
		const uint4 w = *reinterpret_cast<const uint4 *>(tab + 4u * b);
		st = tag_match(w, tt) ? TS_SEEN : ((w.w >> 16) ? TS_FULL : TS_AGAIN);
		if (st == TS_AGAIN)
		{
			// used slots are a prefix: the first word whose upper half is free holds the first free slot
			const bool xo = (w.x >> 16) == 0u, yo = (w.y >> 16) == 0u, zo = (w.z >> 16) == 0u;
			const uint32_t wi  = xo ? 0u : (yo ? 1u : (zo ? 2u : 3u));
			const uint32_t old = xo ? w.x : (yo ? w.y : (zo ? w.z : w.w));
			const uint32_t neu = old ? (old | (tag << 16)) : tag;
			// a failed exchange = another lane's insert changed that word in between: look again (at most 8 rounds)
			if (atomicCAS(tab + 4u * b + wi, old, neu) == old) st = TS_NEW;
		}
	} while (st == TS_AGAIN);
	return st;
}

__device__ __forceinline__ bool tagset_contains(const uint32_t *tab, uint32_t nb, uint32_t magic, uint32_t id)
{
	uint32_t b, tag;
	tagset_split(id, nb, magic, b, tag);
	const uint4 w = *reinterpret_cast<const uint4 *>(tab + 4u * b);
	return tag_match(w, tag * 0x00010001u) != 0u;
}

// The two-set register form was the hot kernel of round 1; the beam form (further down) has dominated it since (40 % fewer VALU
// instructions at 128 dims, 0.76 -> 0.62 ms for one query at 768 dims: profiles/r1i_beam_form.txt) and it had survived only as the
// HNSW_GPU_BEAM=0 fallback and for mirrors of >= 2^31 elements.  Since round 5 those run the generic form below and this kernel is
// compiled in experiment builds only (-DHNSW_EXPERIMENT): 30 instantiations fewer in the shipped library.
#ifdef HNSW_EXPERIMENT
template <int FUNC, typename SH, int RREG>
__global__ __launch_bounds__(256, SH::MIN_WAVES) void hnsw_search_kernel_reg(const SearchArgs a)
{
	constexpr int CREG = 2 * RREG;
	constexpr uint32_t CCAP = 64u * CREG;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float        *qf      = reinterpret_cast<float *>(my);
	const float4 *q4      = reinterpret_cast<const float4 *>(my);
	uint64_t     *tie_key = reinterpret_cast<uint64_t *>(my + a.off_res);     // tie path of the emit only
	uint64_t     *tie_lab = reinterpret_cast<uint64_t *>(my + a.off_cand);    // (both overlay the hash set)
	uint32_t     *htab    = reinterpret_cast<uint32_t *>(my + a.off_hash);
	const uint32_t hmask  = a.hcap - 1;
	uint32_t     *newid   = reinterpret_cast<uint32_t *>(my + a.off_newid);
	float        *newdist = reinterpret_cast<float *>(my + a.off_newdist);

	const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wib;
	uint32_t *vis  = a.vis + (size_t) slot * a.vis_words;
	uint32_t *vlog = a.vlog + (size_t) slot * a.logcap;
	const uint32_t ef = a.ef;
	bool aborted = false;              // the host asked this launch to end (abort word)

	for (;;)
	{
		uint32_t qi = 0;
		if (lane == 0) qi = atomicAdd(a.ticket, 1u);
		qi = __builtin_amdgcn_readfirstlane(qi);
		if (qi >= a.nq) break;
		// (an abort request is sticky for this wave: it takes the remaining tickets without walking and marks every query it does not
		// answer with count 0xFFFFFFFF, so that the caller of an interrupted launch can tell which rows of its outputs are results)
		if (!aborted && (qi & a.abort_mask) == 0u && abort_requested(a)) aborted = true;
		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }   // (wave-uniform by construction; said explicitly)
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi] = __builtin_amdgcn_s_memrealtime();

		const float *qsrc = a.queries + (size_t) qi * a.q_stride;
		for (uint32_t e = lane; e < a.qpad_floats; e += 64)
		{
			const float t = qsrc[e < a.dim ? e : a.dim - 1];
			qf[e] = (e < a.dim) ? t : 0.f;
		}
		wave_sync();
		float qnorm = 0.f;
		if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);

		uint64_t rk[RREG], ck[CREG];
#pragma unroll
		for (int k = 0; k < RREG; k++) rk[k] = ~0ull;
#pragma unroll
		for (int k = 0; k < CREG; k++) ck[k] = ~0ull;
		uint32_t rsize = 0, csize = 0, logn = 0, evals = 0, hops = 0;
		uint32_t hcount = 0;
		bool spill = a.hcap == 0;            // true: visited ids go to the HBM bitmap
		if (a.hcap)
		{
			uint4 *h4 = reinterpret_cast<uint4 *>(htab);
			for (uint32_t i = lane; i < a.hcap / 4; i += 64) h4[i] = make_uint4(HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY);
			wave_sync();
		}

		if (a.n > 0)
		{
			const uint32_t ep = a.entry;                                   // hnswalg.cpp:55-65
			{
				auto one = [ep](uint32_t) { return ep; };
				score_rows<FUNC, SH::KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, one, 1u, newdist, lane);
			}
			wave_sync();
			float lowerBound = finish_dist<FUNC>(newdist[0], newdist[OUT2], qnorm);
			evals = 1;
			if (a.out_evals && a.evals_cap && lane == 0) a.out_evals[(size_t) qi * a.evals_cap] = ep;
			{
				const uint64_t hi = (uint64_t) ord_f32(lowerBound) << 32;
				res_insert<RREG>(rk, hi | ep, lane);
				cand_set<CREG>(ck, 0, hi | (uint32_t) ~ep, lane);
			}
			if (lane == 0)
			{
				if (spill) { vis[ep >> 5] = 1u << (ep & 31); vlog[0] = ep; }
				else htab[hash_slot(ep, hmask)] = ep;
			}
			rsize = csize = 1;
			logn = spill ? 1 : 0;
			hcount = 1;
			wave_sync();

			while (csize > 0)                                               // hnswalg.cpp:67-112
			{
				uint32_t cslot;
				const uint64_t ckey = cand_min<CREG>(ck, cslot);
				if (unord_f32((uint32_t) (ckey >> 32)) > lowerBound)        // :70-71
					break;
				const uint32_t cur = ~(uint32_t) ckey;
				{                                                           // :73 pop = move last into the hole
					const uint32_t last = csize - 1;
					const uint64_t lastkey = res_at<CREG>(ck, last);
					cand_set<CREG>(ck, cslot, lastkey, lane);
					cand_set<CREG>(ck, last, ~0ull, lane);
					csize = last;
				}
				if (a.out_pops && hops < a.pops_cap && lane == 0)       // (system scope: a host that polls the sequence sees it as the walk goes)
					__hip_atomic_store(a.out_pops + (size_t) qi * a.pops_cap + hops, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				hops++;
				if ((hops & 255u) == 0u && abort_requested(a)) { aborted = true; break; }

				for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)               // :76-77
				{
					const uint32_t j = j0 + lane;
					const uint32_t t = a.links[(size_t) cur * a.lstride + (j < a.lstride ? j : a.lstride - 1)];
					bool isnew = false;
					if (j < a.lstride && t != LINK_NONE)                    // :91-93 test-and-set
					{
						if (!spill)
							isnew = hash_test_and_set(htab, hmask, t);
						else if (!(a.hcap && hash_contains(htab, hmask, t)))
						{
							const uint32_t bit = 1u << (t & 31);
							const uint32_t old = atomicOr(&vis[t >> 5], bit);
							isnew = !(old & bit);
						}
					}
					const uint64_t mask = __ballot(isnew);
					const uint32_t nnew = (uint32_t) __builtin_popcountll(mask);
					if (nnew == 0) continue;
					const uint32_t rank = lane_rank(mask);
					if (isnew)
					{
						newid[rank] = t;
						if (a.out_evals && evals + rank < a.evals_cap) a.out_evals[(size_t) qi * a.evals_cap + evals + rank] = t;   // (measurement: the rows this walk scores, in order)
						if (spill)                                          // only bitmap bits need undoing
						{
							const uint32_t lp = logn + rank;
							if (lp < a.logcap) vlog[lp] = t;
						}
					}
					if (spill) logn += nnew;
					else
					{
						hcount += nnew;
						if (hcount + 64 > a.hmax) spill = true;             // keep probe chains short: later ids -> bitmap
					}
					wave_sync();
					{                                                       // :95-97, batched
						const uint32_t *ids = newid;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nnew, newdist, lane);
					}
					evals += nnew;
					wave_sync();
					const float    d_mine = finish_dist<FUNC>(newdist[lane], newdist[OUT2 + lane], qnorm);   // lane r <- row r
					const uint32_t t_mine = newid[lane];
					uint64_t todo = __ballot((uint32_t) lane < nnew && (rsize < ef || lowerBound > d_mine));
					while (todo)                                            // :99-108, in link order
					{
						const uint32_t r = (uint32_t) __builtin_ctzll(todo);
						todo &= todo - 1;
						const float d = __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(d_mine), (int) r));
						if (rsize < ef || lowerBound > d)
						{
							const uint32_t t2 = (uint32_t) __builtin_amdgcn_readlane((int) t_mine, (int) r);
							const uint64_t hi = (uint64_t) ord_f32(d) << 32;
							if (csize == CCAP)                              // make room: the largest key is dead
							{
								uint32_t ms;
								const uint64_t mx = cand_max<CREG>(ck, ms);
								if ((hi | (uint32_t) ~t2) < mx) cand_set<CREG>(ck, ms, hi | (uint32_t) ~t2, lane);
							}
							else
							{
								cand_set<CREG>(ck, csize, hi | (uint32_t) ~t2, lane);   // :100
								csize++;
							}
							res_insert<RREG>(rk, hi | t2, lane);            // :102-105
							rsize = rsize < ef ? rsize + 1 : ef;
							lowerBound = unord_f32((uint32_t) (res_at<RREG>(rk, rsize - 1) >> 32));   // :107
						}
					}
					wave_sync();
				}
			}
		}

		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }      // interrupted inside its walk
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi + 1] = __builtin_amdgcn_s_memrealtime();
		// ---- emit -------------------------------------------------------------------------
		const size_t obase = (size_t) qi * a.out_stride;
		uint32_t nout = 0;
		if (a.mode == 1)
		{
#pragma unroll
			for (int k = 0; k < RREG; k++)
			{
				const uint32_t i = (uint32_t) k * 64 + lane;
				if (i < ef)
				{
					const bool ok = i < rsize;
					a.out_idx[obase + i] = ok ? (uint32_t) rk[k] : LINK_NONE;
					if (a.out_dists) a.out_dists[obase + i] = ok ? unord_f32((uint32_t) (rk[k] >> 32)) : __builtin_inff();
				}
			}
			for (uint32_t i = ef + lane; i < a.out_stride; i += 64)     // caller's ef larger than the index
			{
				a.out_idx[obase + i] = LINK_NONE;
				if (a.out_dists) a.out_dists[obase + i] = __builtin_inff();
			}
			nout = rsize;
		}
		else
		{
			// searchKnn, hnswalg.cpp:241-249
			uint64_t lab[RREG];
			bool tie = false;
#pragma unroll
			for (int k = 0; k < RREG; k++)
			{
				const uint32_t i = (uint32_t) k * 64 + lane;
				lab[k] = a.labels[(i < rsize) ? (uint32_t) rk[k] : 0];
				const uint64_t fill = (k > 0) ? readlane_u64(rk[k > 0 ? k - 1 : 0], 63) : 0ull;
				const uint64_t prev = wave_shr1_u64(rk[k], fill);
				if (i > 0 && i < rsize && (uint32_t) (prev >> 32) == (uint32_t) (rk[k] >> 32)) tie = true;
			}
			if (__ballot(tie) == 0)
			{
#pragma unroll
				for (int k = 0; k < RREG; k++)
				{
					const uint32_t i = (uint32_t) k * 64 + lane;
					const bool keep = i < rsize && !((lab[k] >> 48) & 1);
					const uint64_t kmask = __ballot(keep);
					if (keep)
					{
						const uint32_t rank = nout + lane_rank(kmask);
						a.out_labels[obase + rank] = lab[k];
						if (a.out_dists) a.out_dists[obase + rank] = unord_f32((uint32_t) (rk[k] >> 32));
					}
					nout += (uint32_t) __builtin_popcountll(kmask);
				}
			}
			else
			{
				// equal distances present: order by (dist, label) through LDS (hnswalg.cpp:236,246)
#pragma unroll
				for (int k = 0; k < RREG; k++)
				{
					const uint32_t i = (uint32_t) k * 64 + lane;
					if (i < rsize) { tie_key[i] = rk[k]; tie_lab[i] = lab[k]; }
				}
				wave_sync();
#pragma unroll
				for (int k = 0; k < RREG; k++)
				{
					const uint32_t i = (uint32_t) k * 64 + lane;
					const bool keep = i < rsize && !((lab[k] >> 48) & 1);
					const uint32_t di = (uint32_t) (rk[k] >> 32);
					uint32_t rank = 0;
					for (uint32_t jx = 0; jx < rsize; jx++)
					{
						const uint64_t lj = tie_lab[jx];
						const uint32_t dj = (uint32_t) (tie_key[jx] >> 32);
						const bool kj = !((lj >> 48) & 1);
						rank += (kj && (dj < di || (dj == di && lj < lab[k]))) ? 1u : 0u;
					}
					if (keep)
					{
						a.out_labels[obase + rank] = lab[k];
						if (a.out_dists) a.out_dists[obase + rank] = unord_f32(di);
					}
					nout += (uint32_t) __builtin_popcountll(__ballot(keep));
				}
			}
			for (uint32_t i = nout + lane; i < a.out_stride; i += 64)
			{
				a.out_labels[obase + i] = ~0ull;
				if (a.out_dists) a.out_dists[obase + i] = __builtin_inff();
			}
		}
		if (lane == 0)
		{
			a.out_counts[qi] = nout;
			if (a.out_stats) { a.out_stats[2 * (size_t) qi] = evals; a.out_stats[2 * (size_t) qi + 1] = hops; }
		}
		if (a.done) signal_done(a.done + qi, lane);

		// ---- restore the all-zero bitmap ---------------------------------------------------
		wave_sync();
		if (logn <= a.logcap)
		{
			for (uint32_t i = lane; i < logn; i += 64) vis[vlog[i] >> 5] = 0u;
		}
		else
		{
			for (uint64_t w = lane; w < a.vis_words; w += 64) vis[w] = 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0);
		wave_sync();
	}
	if (aborted && lane == 0) atomicAdd(a.health + HEALTH_ABORTED_WAVES, 1u);
}


#endif      // HNSW_EXPERIMENT (two-set register form)

// =====================================================================================
// Generic form (any ef; used when ef > 256): both sets as UNSORTED arrays in LDS.
//   results    : res[0..rsize) + the position of the largest key kept wave-uniformly.  Insert while
//                not full = append; when full = overwrite the largest and rescan for the new largest
//                (ceil(ef/64) LDS reads per lane + one DPP wave-min).
//   candidates : cand[0..csize), capacity 2*ef (exact, see the header).  Append = one LDS write;
//                pop-best = scan for the smallest key, move the last entry into the hole.
//   emit       : rank sort by (dist, idx) or (dist, label) — O(ef^2/64) per query, a few percent
//                of a traversal that long.
// Visited set = the per-slot HBM bitmap.  Same pre-filtered accept loop as the register form.
// =====================================================================================

// Set-array accessors.  G = false: LDS, plain accesses.  G = true: HBM scratch, accessed with relaxed
// agent-scope atomics = L1-bypassing loads/stores, so that a wave always reads back its own writes from L2
// (loads still pipeline: the scans below issue four before the first use).
template <bool G>
__device__ __forceinline__ uint64_t ldk(const uint64_t *p)
{
	if (G) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}
template <bool G>
__device__ __forceinline__ void stk(uint64_t *p, uint64_t v)
{
	if (G) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else *p = v;
}

// Smallest (MIN=true) or largest key of A[0..n) and its position; n > 0; wave-uniform result.
template <bool MIN, bool G>
__device__ __forceinline__ uint64_t lds_extreme(const uint64_t *A, uint32_t n, uint32_t &pos, int lane)
{
	uint64_t best = MIN ? ~0ull : 0ull;
	uint32_t bpos = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += 256)
	{
		uint64_t k[4];
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const uint32_t i = i0 + 64u * u + lane;
			k[u] = ldk<G>(&A[i < n ? i : n - 1]);
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const uint32_t i = i0 + 64u * u + lane;
			const bool better = i < n && (MIN ? (k[u] < best) : (k[u] > best));
			best = better ? k[u] : best;
			bpos = better ? i : bpos;
		}
	}
	// reduce on the distance word, then on the low word among the lanes that tie on it
	const uint32_t h = MIN ? (uint32_t) (best >> 32) : ~(uint32_t) (best >> 32);
	const uint32_t hmin = wave_min_u32(h);
	uint64_t eq = __ballot(h == hmin);
	if (__builtin_popcountll(eq) > 1)
	{
		const uint32_t lo = (h == hmin) ? (MIN ? (uint32_t) best : ~(uint32_t) best) : 0xFFFFFFFFu;
		const uint32_t lomin = wave_min_u32(lo);
		eq = __ballot(h == hmin && lo == lomin);
	}
	const uint32_t L = (uint32_t) __builtin_ctzll(eq);
	pos = (uint32_t) __builtin_amdgcn_readlane((int) bpos, (int) L);
	return readlane_u64(best, L);
}

// Make this wave's own writes to the set arrays visible to its own later reads.  LDS: program order +
// lgkmcnt.  HBM (G): the accesses bypass L1 (ldk/stk), so draining vmcnt is enough.
template <bool G>
__device__ __forceinline__ void set_sync()
{
	if (G)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0);
	}
	wave_sync();
}

// G = false: result/candidate arrays in LDS (ef up to what 160 KB hold).  G = true: the same arrays in a
// per-slot HBM scratch area — any ef the API admits (the reference's scan doubles efSearch until the
// index is exhausted, embedding.c:329-343), at L2 latency per scan instead of LDS latency.
template <int FUNC, typename SH, bool G>
__global__ __launch_bounds__(256) void hnsw_search_kernel_lds(const SearchArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float        *qf      = reinterpret_cast<float *>(my);
	const float4 *q4      = reinterpret_cast<const float4 *>(my);
	uint32_t     *newid   = reinterpret_cast<uint32_t *>(my + a.off_newid);
	float        *newdist = reinterpret_cast<float *>(my + a.off_newdist);

	const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wib;
	uint64_t *res, *cand;
	if (G)
	{
		res  = a.set_scratch + (size_t) slot * a.set_stride;
		cand = res + a.off_cand;                        // G: off_cand counts keys inside the slot's area
	}
	else
	{
		res  = reinterpret_cast<uint64_t *>(my + a.off_res);
		cand = reinterpret_cast<uint64_t *>(my + a.off_cand);
	}
	uint32_t *vis  = a.vis + (size_t) slot * a.vis_words;
	uint32_t *vlog = a.vlog + (size_t) slot * a.logcap;
	const uint32_t ef = a.ef;
	bool aborted = false;              // the host asked this launch to end (abort word)

	for (;;)
	{
		uint32_t qi = 0;
		if (lane == 0) qi = atomicAdd(a.ticket, 1u);
		qi = __builtin_amdgcn_readfirstlane(qi);
		if (qi >= a.nq) break;
		// (an abort request is sticky for this wave: it takes the remaining tickets without walking and marks every query it does not
		// answer with count 0xFFFFFFFF, so that the caller of an interrupted launch can tell which rows of its outputs are results)
		if (!aborted && (qi & a.abort_mask) == 0u && abort_requested(a)) aborted = true;
		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }   // (wave-uniform by construction; said explicitly)
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi] = __builtin_amdgcn_s_memrealtime();

		const float *qsrc = a.queries + (size_t) qi * a.q_stride;
		for (uint32_t e = lane; e < a.qpad_floats; e += 64)
		{
			const float t = qsrc[e < a.dim ? e : a.dim - 1];     // unconditional load, then select
			qf[e] = (e < a.dim) ? t : 0.f;
		}
		set_sync<G>();
		float qnorm = 0.f;
		if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);

		uint32_t rsize = 0, csize = 0, logn = 0, evals = 0, hops = 0;
		uint32_t rmax_pos = 0;

		if (a.n > 0)      // empty index: hnsw_begin_read(entry) fails, hnswalg.cpp:56-57
		{
			const uint32_t ep = a.entry;                                   // hnswalg.cpp:55-65
			{
				auto one = [ep](uint32_t) { return ep; };
				score_rows<FUNC, SH::KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, one, 1u, newdist, lane);
			}
			set_sync<G>();
			float lowerBound = finish_dist<FUNC>(newdist[0], newdist[OUT2], qnorm);
			evals = 1;
			if (a.out_evals && a.evals_cap && lane == 0) a.out_evals[(size_t) qi * a.evals_cap] = ep;
			if (lane == 0)
			{
				const uint32_t o = ord_f32(lowerBound);
				stk<G>(&res[0], ((uint64_t) o << 32) | ep);
				stk<G>(&cand[0], ((uint64_t) o << 32) | (uint32_t) ~ep);
				vis[ep >> 5] = 1u << (ep & 31);       // slot bitmap is all-zero here
				vlog[0] = ep;
			}
			rsize = csize = logn = 1;
			set_sync<G>();

			while (csize > 0)                                               // hnswalg.cpp:67-112
			{
				uint32_t cpos;
				const uint64_t ck = lds_extreme<true, G>(cand, csize, cpos, lane);
				if (unord_f32((uint32_t) (ck >> 32)) > lowerBound)         // :70-71
					break;
				const uint32_t cur = ~(uint32_t) ck;
				csize--;                                                    // :73 pop = last entry into the hole
				if (lane == 0) stk<G>(&cand[cpos], ldk<G>(&cand[csize]));
				set_sync<G>();
				if (a.out_pops && hops < a.pops_cap && lane == 0)       // (system scope: a host that polls the sequence sees it as the walk goes)
					__hip_atomic_store(a.out_pops + (size_t) qi * a.pops_cap + hops, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				hops++;
				if ((hops & 255u) == 0u && abort_requested(a)) { aborted = true; break; }

				for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)               // :76-77
				{
					const uint32_t j = j0 + lane;
					const uint32_t t = a.links[(size_t) cur * a.lstride + (j < a.lstride ? j : a.lstride - 1)];
					bool isnew = false;
					if (j < a.lstride && t != LINK_NONE)                    // :91-93 test-and-set
					{
						const uint32_t bit = 1u << (t & 31);
						const uint32_t old = atomicOr(&vis[t >> 5], bit);
						isnew = !(old & bit);
					}
					const uint64_t mask = __ballot(isnew);
					const uint32_t nnew = (uint32_t) __builtin_popcountll(mask);
					if (nnew == 0) continue;
					const uint32_t rank = lane_rank(mask);
					if (isnew)
					{
						newid[rank] = t;
						if (a.out_evals && evals + rank < a.evals_cap) a.out_evals[(size_t) qi * a.evals_cap + evals + rank] = t;   // (measurement: the rows this walk scores, in order)
						const uint32_t lp = logn + rank;
						if (lp < a.logcap) vlog[lp] = t;
					}
					logn += nnew;
					set_sync<G>();
					{                                                       // :95-97, batched
						const uint32_t *ids = newid;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nnew, newdist, lane);
					}
					evals += nnew;
					set_sync<G>();
					const float    d_mine = finish_dist<FUNC>(newdist[lane], newdist[OUT2 + lane], qnorm);
					const uint32_t t_mine = newid[lane];
					uint64_t todo = __ballot((uint32_t) lane < nnew && (rsize < ef || lowerBound > d_mine));
					while (todo)                                            // :99-108, in link order
					{
						const uint32_t r = (uint32_t) __builtin_ctzll(todo);
						todo &= todo - 1;
						const float d = __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(d_mine), (int) r));
						if (!(rsize < ef || lowerBound > d)) continue;
						const uint32_t t2 = (uint32_t) __builtin_amdgcn_readlane((int) t_mine, (int) r);
						const uint64_t hi = (uint64_t) ord_f32(d) << 32;
						const uint64_t ckey = hi | (uint32_t) ~t2, rkey = hi | t2;
						if (csize == a.ccap)                                // :100; make room: the largest key is dead
						{
							uint32_t mp;
							const uint64_t mx = lds_extreme<false, G>(cand, csize, mp, lane);
							if (ckey < mx && lane == 0) stk<G>(&cand[mp], ckey);
						}
						else
						{
							if (lane == 0) stk<G>(&cand[csize], ckey);
							csize++;
						}
						if (rsize < ef)                                     // :102
						{
							if (lane == 0) stk<G>(&res[rsize], rkey);
							rsize++;
							set_sync<G>();
							if (rsize == 1 || rkey > ldk<G>(&res[rmax_pos])) rmax_pos = rsize - 1;
						}
						else                                                // :104-105 evict the largest
						{
							if (lane == 0) stk<G>(&res[rmax_pos], rkey);
							set_sync<G>();
							(void) lds_extreme<false, G>(res, rsize, rmax_pos, lane);
						}
						set_sync<G>();
						lowerBound = unord_f32((uint32_t) (ldk<G>(&res[rmax_pos]) >> 32));   // :107
					}
					set_sync<G>();
				}
			}
		}

		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }      // interrupted inside its walk
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi + 1] = __builtin_amdgcn_s_memrealtime();
		// ---- emit: rank-sort the unsorted result array ----------------------------------------
		// (G: the arrays are final now; drop this CU's L1 copies of them once — an earlier query of this slot
		// read them through L1 here — and read them with plain, freely pipelined loads)
		if (G) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		const size_t obase = (size_t) qi * a.out_stride;
		uint32_t nout = 0;
		if (a.mode == 1)
		{
			for (uint32_t b = 0; b < rsize; b += 64)
			{
				const uint32_t i = b + lane;
				if (i < rsize)
				{
					const uint64_t k = res[i];
					uint32_t rank = 0;
					for (uint32_t jx = 0; jx < rsize; jx++) rank += (res[jx] < k) ? 1u : 0u;
					a.out_idx[obase + rank] = (uint32_t) k;
					if (a.out_dists) a.out_dists[obase + rank] = unord_f32((uint32_t) (k >> 32));
				}
			}
			nout = rsize;
			for (uint32_t i = nout + lane; i < a.out_stride; i += 64)
			{
				a.out_idx[obase + i] = LINK_NONE;
				if (a.out_dists) a.out_dists[obase + i] = __builtin_inff();
			}
		}
		else
		{
			// searchKnn, hnswalg.cpp:241-249: label lookup, vacuum filter, order by (dist, label)
			uint64_t *lab = cand;                       // candidate array is dead now (capacity 2*ef)
			for (uint32_t i = lane; i < rsize; i += 64) lab[i] = a.labels[(uint32_t) res[i]];
			set_sync<G>();
			if (G) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // lab[] is read through L1 below
			for (uint32_t b = 0; b < rsize; b += 64)
			{
				const uint32_t i = b + lane;
				const bool in = i < rsize;
				const uint64_t li = in ? lab[i] : 0;
				const uint32_t di = in ? (uint32_t) (res[i] >> 32) : 0;
				const bool keep = in && !((li >> 48) & 1);           // hnsw_is_deleted, embedding.c:948-953
				uint32_t rank = 0;
				for (uint32_t jx = 0; jx < rsize; jx++)              // rank by (dist, label), hnswalg.cpp:236,246
				{
					const uint64_t lj = lab[jx];
					const uint32_t dj = (uint32_t) (res[jx] >> 32);
					const bool kj = !((lj >> 48) & 1);
					rank += (kj && (dj < di || (dj == di && lj < li))) ? 1u : 0u;
				}
				if (keep)
				{
					a.out_labels[obase + rank] = li;
					if (a.out_dists) a.out_dists[obase + rank] = unord_f32(di);
				}
				nout += (uint32_t) __builtin_popcountll(__ballot(keep));
			}
			for (uint32_t i = nout + lane; i < a.out_stride; i += 64)          // pad the tail
			{
				a.out_labels[obase + i] = ~0ull;
				if (a.out_dists) a.out_dists[obase + i] = __builtin_inff();
			}
		}
		if (lane == 0)
		{
			a.out_counts[qi] = nout;
			if (a.out_stats) { a.out_stats[2 * (size_t) qi] = evals; a.out_stats[2 * (size_t) qi + 1] = hops; }
		}
		if (a.done) signal_done(a.done + qi, lane);

		// ---- restore the all-zero bitmap for the next query of this slot --------------
		set_sync<G>();
		if (logn <= a.logcap)
		{
			for (uint32_t i = lane; i < logn; i += 64) vis[vlog[i] >> 5] = 0u;
		}
		else
		{
			for (uint64_t w = lane; w < a.vis_words; w += 64) vis[w] = 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0);   // drain: the next query's atomics must see the zeros
		set_sync<G>();
	}
	if (aborted && lane == 0) atomicAdd(a.health + HEALTH_ABORTED_WAVES, 1u);
}


// =====================================================================================
// Beam form (ef <= 64*UREG/2): ONE unsorted set of accepted elements, acceptance by counting.
//
// The reference keeps topResults and candidateSet; both are functions of the SET of accepted
// elements, and every decision it takes can be restated as a count over that set:
//   * accept x   (hnswalg.cpp:99: top().first > dist || size < ef)
//         <=>  fewer than ef accepted elements have dist <= dist(x);
//   * stop       (hnswalg.cpp:70: best candidate > lowerBound)
//         <=>  at least ef accepted elements have dist < dist(best unexpanded element);
//   * topResults at the end = the ef smallest (dist, idx) keys of the accepted set;
//   * candidateSet.top() = the unexpanded accepted element with the smallest (dist, ~idx) key.
// (The bound only shrinks, so an element rejected or evicted earlier never counts against a later
// acceptance: it lies at or above the bound of its time, hence above any later accepted distance.
// The same argument lets the set be PRUNED to the elements at or below the ef-th smallest distance —
// ties at that distance are kept, so both counts and the live candidates stay exact.)
// So the kernel keeps `uk[UREG]` = accepted keys ord(dist)<<32 | idx, unsorted, one "expanded" bit per
// slot, and NO sorted insert and NO second set:
//   accept test = UREG compares + ballots; append = one slot write; pop = masked min-scan;
//   prune (when the 64*UREG slots are full): 32-step radix select of the ef-th smallest distance word,
//   compaction in registers (ds_permute; rounds 1-5: through a per-slot HBM scratch line), once per ~ef accepts.
// Output order is produced at the end by a rank sort over the <= ef survivors.
// =====================================================================================

template <int U>
__device__ __forceinline__ uint32_t beam_count_le(const uint64_t (&uk)[U], uint32_t od)
{
	uint32_t c = 0;
#pragma unroll
	for (int k = 0; k < U; k++) c += (uint32_t) __builtin_popcountll(__ballot((uint32_t) (uk[k] >> 32) <= od));
	return c;
}

template <int U>
__device__ __forceinline__ uint32_t beam_count_lt(const uint64_t (&uk)[U], uint32_t od)
{
	uint32_t c = 0;
#pragma unroll
	for (int k = 0; k < U; k++) c += (uint32_t) __builtin_popcountll(__ballot((uint32_t) (uk[k] >> 32) < od));
	return c;
}

// Best unexpanded element: smallest (dist, ~idx).  Returns false when none is left.
// The distance word decides alone unless two open elements share the smallest one (then the larger idx goes first,
// as std::pair<-dist, idx> orders them): per-lane minimum of the open distance words, one wave minimum, and the slot
// from UREG equality ballots — no 64-bit compares, no carried (key, register) pairs.
// (WITH_LT: also the stop test's count — how many elements of the set, open or not, lie below the popped distance (hnswalg.cpp:70) — taken here,
// where its compares issue together with the equality ballots instead of behind the slot selection.  Only for a pop that is used at once: the
// early scan of the one-wave kernels looks at a set that the hop's accepts still change.)
template <int U, bool WITH_LT = false>
__device__ __forceinline__ bool beam_next(const uint64_t (&uk)[U], uint32_t ex, uint32_t &slot, uint64_t &ckey, uint32_t *n_lt = nullptr)
{
	uint32_t h[U];
	uint32_t m = 0xFFFFFFFFu;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		h[k] = ((ex >> k) & 1u) ? 0xFFFFFFFFu : (uint32_t) (uk[k] >> 32);      // unused slots hold ~0 already
		m = min(m, h[k]);
	}
	const uint32_t hmin = wave_min_u32(m);
	if (hmin == 0xFFFFFFFFu) return false;
	uint64_t eq[U];
	uint32_t cnt = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		eq[k] = __ballot(h[k] == hmin);
		cnt += (uint32_t) __builtin_popcountll(eq[k]);
	}
	if (WITH_LT)
	{
		uint32_t c = 0;
#pragma unroll
		for (int k = 0; k < U; k++) c += (uint32_t) __builtin_popcountll(__ballot((uint32_t) (uk[k] >> 32) < hmin));
		*n_lt = c;
	}
	if (cnt > 1)                                            // equal distances: larger idx first
	{
		uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
		for (int k = 0; k < U; k++) lo = min(lo, h[k] == hmin ? ~(uint32_t) uk[k] : 0xFFFFFFFFu);
		const uint32_t lomin = wave_min_u32(lo);
#pragma unroll
		for (int k = 0; k < U; k++) eq[k] = __ballot(h[k] == hmin && ~(uint32_t) uk[k] == lomin);
	}
	uint32_t ks = U - 1;
	uint64_t em = eq[U - 1];
#pragma unroll
	for (int k = U - 2; k >= 0; k--)
	{
		ks = eq[k] ? (uint32_t) k : ks;
		em = eq[k] ? eq[k] : em;
	}
	const uint32_t L = (uint32_t) __builtin_ctzll(em);
	uint32_t idx = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const uint32_t t = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) uk[k], (int) L);
		idx = ks == (uint32_t) k ? t : idx;
	}
	slot = (ks << 6) | L;
	ckey = ((uint64_t) hmin << 32) | (uint32_t) ~idx;
	return true;
}

// bitwise OR over the wavefront, returned uniformly (the DPP steps of wave_min_u32 with `|`; identity 0)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
	v |= dpp_u32<0xB1>(0u, v);
	v |= dpp_u32<0x4E>(0u, v);
	v |= dpp_u32<0x141>(0u, v);
	v |= dpp_u32<0x140>(0u, v);
	v |= dpp_u32<0x142, 0xA>(0u, v);
	v |= dpp_u32<0x143, 0xC>(0u, v);
	return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

// ef-th smallest distance word of the set (set holds >= ef used slots).
// (The set holds the best ~ef .. 2 ef elements of a walk: their distance words share the sign, the exponent or all but its lowest bits, often
// some of the mantissa.  The bits ALL used slots share decide nothing, so the radix descent starts below them — one OR-reduction of the
// differences against slot 0 buys ~8-10 of the 32 steps.)
template <int U>
__device__ __forceinline__ uint32_t beam_select(const uint64_t (&uk)[U], uint32_t ef)
{
#ifdef HNSW_OLD_SELECT
	int top = 31;
	uint32_t prefix = 0, need = ef;
#else
	const uint32_t ref = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (uk[0] >> 32), 0);     // slot 0 is always in use
	uint32_t diff = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const uint32_t h = (uint32_t) (uk[k] >> 32);
		diff |= h != 0xFFFFFFFFu ? h ^ ref : 0u;
	}
	diff = wave_or_u32(diff);
	if (diff == 0u) return ref;                                     // every used slot holds the same distance word
	const int top = 31 - __builtin_clz(diff);
	uint32_t prefix = top == 31 ? 0u : ref & ~((2u << top) - 1u), need = ef;
#endif
	for (int bit = top; bit >= 0; bit--)
	{
		uint32_t c = 0;
#pragma unroll
		for (int k = 0; k < U; k++)
			c += (uint32_t) __builtin_popcountll(__ballot(((((uint32_t) (uk[k] >> 32)) ^ prefix) >> bit) == 0));
		if (c < need) { need -= c; prefix |= 1u << bit; }
	}
	return prefix;
}

// Drop every element whose distance word exceeds `v`; survivors keep their expanded bits and are
// compacted to slots 0..n-1.  Returns n.
#ifndef HNSW_OLD_COMPACT
// (round 6: in registers.  Register k's survivors go to slots base .. base + n - 1: every lane pushes its key with ds_permute — a survivor to the
// lane that owns its slot, the others to the lanes behind, a permutation — and a lane keeps what it received if that is one of the n survivors, in
// the register its slot belongs to (two candidates per source register).  Rounds 1-5 went through a per-slot HBM scratch line: two memory round
// trips per prune, ~5 prunes per walk.  `scratch` is unused.)
template <int U>
__device__ __forceinline__ uint32_t beam_compact(uint64_t (&uk)[U], uint32_t &ex, uint32_t v, uint64_t *scratch, int lane)
{
	(void) scratch;
	uint64_t nk[U];
#pragma unroll
	for (int k = 0; k < U; k++) nk[k] = ~0ull;
	uint32_t base = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const uint32_t hi = (uint32_t) (uk[k] >> 32);
		const bool keep = hi <= v && hi != 0xFFFFFFFFu;
		const uint64_t mask = __ballot(keep);
		const uint32_t n = (uint32_t) __builtin_popcountll(mask);
		if (n)                                                      // (wave-uniform)
		{
			const uint32_t rank = lane_rank(mask);
			const uint32_t dst = base + (keep ? rank : n + ((uint32_t) lane - rank));
			// expanded bit travels in bit 31 of the idx word (element numbers stay below 2^31 in this form)
			const uint32_t lo = (uint32_t) uk[k] | (((ex >> k) & 1u) << 31);
			const uint32_t r_hi = (uint32_t) __builtin_amdgcn_ds_permute((int) ((dst & 63u) << 2), (int) hi);
			const uint32_t r_lo = (uint32_t) __builtin_amdgcn_ds_permute((int) ((dst & 63u) << 2), (int) lo);
			const uint32_t pos = ((uint32_t) lane - base) & 63u;
			const uint32_t dreg = pos < n ? (base + pos) >> 6 : 0xFFFFFFFFu;
#pragma unroll
			for (int d = 0; d < U; d++) nk[d] = dreg == (uint32_t) d ? (((uint64_t) r_hi << 32) | r_lo) : nk[d];
		}
		base += n;
	}
	ex = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const bool used = (uint32_t) k * 64 + (uint32_t) lane < base;
		ex |= (used && ((nk[k] >> 31) & 1u)) ? (1u << k) : 0u;
		uk[k] = used ? (nk[k] & ~(1ull << 31)) : ~0ull;
	}
	return base;
}
#else
template <int U>
__device__ __forceinline__ uint32_t beam_compact(uint64_t (&uk)[U], uint32_t &ex, uint32_t v, uint64_t *scratch, int lane)
{
	uint32_t base = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const bool keep = (uint32_t) (uk[k] >> 32) <= v && (uint32_t) (uk[k] >> 32) != 0xFFFFFFFFu;
		const uint64_t mask = __ballot(keep);
		// expanded bit travels in bit 31 of the idx word (element numbers stay below 2^31 in this form)
		if (keep) scratch[base + lane_rank(mask)] = uk[k] | ((uint64_t) ((ex >> k) & 1u) << 31);
		base += (uint32_t) __builtin_popcountll(mask);
	}
	// same wave, same L2: the stores are acknowledged by L2 once vmcnt drains, and the loads below
	// bypass L1 — no cache writeback needed (an agent-scope release costs a full L2 writeback on gfx950)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_s_waitcnt(0);
	ex = 0;
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const uint32_t i = (uint32_t) k * 64 + lane;
		uint64_t t = ~0ull;
		if (i < base) t = __hip_atomic_load(&scratch[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L1-bypassing load
		const bool used = i < base;
		ex |= (used && ((t >> 31) & 1u)) ? (1u << k) : 0u;
		uk[k] = used ? (t & ~(1ull << 31)) : ~0ull;
	}
	return base;
}
#endif

template <int U>
__device__ __forceinline__ void beam_set(uint64_t (&uk)[U], uint32_t slot, uint64_t key, int lane)
{
#pragma unroll
	for (int k = 0; k < U; k++)
	{
		const bool hit = slot == ((uint32_t) k * 64 + (uint32_t) lane);
		uk[k] = hit ? key : uk[k];
	}
}


// =====================================================================================
// Team form (TEAM = true): waves of a block that have no query left help a sibling's walk.
//
// One query is a chain of ~ef dependent hops, each two memory round trips (link list, then rows) plus the
// set upkeep: 0.6-0.7 ms at 768 dims however idle the chip is — the shape of the reference's own call
// (one query per hnsw_search, embedding.c:317) and of the tail of every batch (its last, longest walks run
// on an empty chip).  The walk cannot be reordered without changing results, but its INPUTS can be fetched
// ahead: a distance is a pure function of (query, row), a link list a pure function of the element.
//   * The walking wave ("main") publishes its accepted set (keys + expanded bits) once per hop.
//   * A helper picks the r-th best unexpanded element of that snapshot (r = its rank among the helpers; r = 0
//     is the predicted next pop: right in 83-85 % of the hops), loads its link list, scores the neighbours the
//     main has not visited yet with the SAME canonical code, and leaves a "package" for that element in a slot of
//     its own LDS region (it has no query of its own, so helpers cost no extra LDS and write only their own
//     region): per link position (neighbour id, ord(dist)).
//   * The main walks exactly as before — pops, marks visited and accepts in the reference's order — but when a
//     package for the popped element exists it takes link list and distances from it (waiting for a helper that
//     has the element in flight: that helper is further along than a fresh fetch would be and depends on
//     nothing the main does) and fetches/scores only what a package does not hold.
// Exactness: a packaged value is identical, bit for bit, to what the main would compute (same code, same
// summation order); a package is validated by re-reading its header (element, state) after the copy; a
// neighbour a helper skipped as visited IS visited (the main's visited set only grows during a walk), and
// nothing a helper does can change WHICH values the walk consumes or in what order.  A stale or torn snapshot
// only makes a helper fetch something useless.  Outputs, E_q and H_q equal the one-wave form
// (tests/test_gpu_search.py::test_every_kernel_variant_is_exact, ::test_team_form_is_exact_at_every_launch_size).
// =====================================================================================
// launch-wide diagnostics of the team form (hnsw_gpu_team_counters): compiled in only with -DHNSW_TEAM_COUNTERS —
// the cycle stamps and global atomics cost ~15 % of a single-query walk
#ifdef HNSW_TEAM_COUNTERS
constexpr bool TEAM_COUNT = true;
#else
constexpr bool TEAM_COUNT = false;
#endif
// per-section cycle stamps of the walking wave (any beam kernel): compiled in only with -DHNSW_HOP_STAMPS
// (scripts/build_variant.sh); sums in units of 64 cycles land in team_dbg[0..7]
#ifdef HNSW_HOP_STAMPS
constexpr bool HOP_STAMPS = true;
#else
constexpr bool HOP_STAMPS = false;
#endif
__device__ __forceinline__ uint32_t hop_stamp()
{
	__builtin_amdgcn_sched_barrier(0);
	const uint32_t t = (uint32_t) __builtin_amdgcn_s_memtime();
	__builtin_amdgcn_sched_barrier(0);
	return t;
}

constexpr uint32_t SLICE_ROWS = 16;             // rows of the widest scoring pass (4 * RPG, RPG <= 4) = the largest slice
struct TeamCtl
{
	uint32_t state;        // 0 = has or may get a query but is not walking, 1 = walking, 2 = will never walk again
	uint32_t helpers;      // main: bit h set = wave h of this block is helping me
	uint32_t job;          // main: a scoring job for its slice helpers: view << 15 | helper mask << 7 | rows ...
	uint32_t jobseq;       // ... and its number (written after `job`, read before and after it: a helper never pairs one
	                       //     job's number with another job's description)
	uint32_t done;         // helper: the last job whose slice I finished (my sums are in `slice`): job number * 8 + the walking
	                       //     wave it was for — a helper moves from walk to walk, and two walking waves' job numbers may coincide
	uint32_t pad0, pad1, pad2;
	float    slice[2 * SLICE_ROWS];     // helper: sums of my slice [0, SLICE_ROWS) and, cosine, |x|^2 behind them
};
// A hop of the descent from the entry point brings 20-32 unvisited rows and no package can exist for it (the popped
// element was accepted one hop earlier): 3-4 scoring passes of the walking wave alone, each a full HBM round trip —
// a tenth of a single query's time (profiles/r2zf_team_hop_pattern.txt).  Helpers beyond the first tm_spec do not
// speculate; they wait for such a hop and score 8-row slices of it with the same canonical code, so the hop costs ONE
// round trip.  Protocol (every wait in it is BOUNDED; what a helper does not deliver in time the walking wave scores
// itself, so a protocol mistake costs time, never a hang and never a wrong sum):
//   main    writes the id list (its view's `miss` array), then ctl[main].job, then ++ctl[main].jobseq; scores the first
//           pass itself; then, per named helper, polls ctl[helper].done == jobseq (bounded), copies that helper's
//           `slice` into its own sum array — or, after the bound, scores the slice itself;
//   helper  sees jobseq change (it remembered the number from BEFORE its helper bit became visible, so a job that names
//           it always looks new), reads job between two reads of jobseq, scores its slice into ITS OWN ctl[me].slice
//           (never into the main's arrays: a helper that is late for a job the main gave up on writes where nobody
//           reads), then stores ctl[me].done = that job's number * 8 + the main's wave number.  A late `done` never equals
//           a later job's value, and — a helper moves from walk to walk, and the job numbers of two walking waves of a
//           block may coincide — never another main's either (found on the device at Q = 1024 / 1536 dims, where helpers
//           change walks all the time: profiles/r3d_c5_spec_mismatch.txt; it is also reset when a helper attaches).
// (job word: the helper mask is the walking wave's own snapshot, so a helper that attaches meanwhile changes nobody's
// slice; `view` = whose region holds the id list.)
constexpr uint32_t SLICE_WAIT_POLLS = 20000;    // ~1 ms of polling: two orders of magnitude above a slice's round trip
__device__ __forceinline__ uint32_t lds_load_u32(const uint32_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr uint64_t DC_EMPTY = ~0ull;
constexpr uint32_t OD_MISSING = 0xFFFFFFFFu;      // package entry without a distance (ord() of a real distance is never all ones: that is a NaN payload no sum produces... guarded anyway: such an entry is simply re-scored)
constexpr uint32_t PK_VISITED = 0x80000000u;
constexpr int TEAM_MAX_WPB = 8;
enum : uint32_t { LC_CLAIMED = 1u, LC_DONE = 2u };   // package slot: claimed (in flight) -> complete

struct TeamView            // the carve of one (helper's) LDS region
{
	uint64_t *pub;         // 64*UREG keys: the main's accepted set (only in the region of its lowest helper)
	uint32_t *ex;          // 64 words: expanded bits per lane
	uint32_t *miss;        // 64 ids: the main's compaction scratch for neighbours a package has no distance for
	uint64_t *hdr;         // tm_lcslots direct-mapped package slots: element | state << 32 ...
	uint64_t *pk;          // ... and lstride entries each: neighbour id | ord(dist) << 32
	uint64_t *dc;          // the helper's own memo of distances it has computed for this walk: id << 32 | ord(dist)
};

__device__ __forceinline__ TeamView team_view(unsigned char *smem, const SearchArgs &a, uint32_t w)
{
	unsigned char *r = smem + (size_t) w * a.wave_bytes;
	TeamView v;
	v.pub  = reinterpret_cast<uint64_t *>(r);
	v.ex   = reinterpret_cast<uint32_t *>(r + a.tm_off_ex);
	v.miss = reinterpret_cast<uint32_t *>(r + a.tm_off_miss);
	v.hdr  = reinterpret_cast<uint64_t *>(r + a.tm_off_lctag);
	v.pk   = reinterpret_cast<uint64_t *>(r + a.tm_off_lclinks);
	v.dc   = reinterpret_cast<uint64_t *>(r + a.tm_off_dc);
	return v;
}

__device__ __forceinline__ uint32_t lc_slot(uint32_t id, uint32_t slots)
{
	return ((id * 0x9E3779B1u) >> 11) & (slots - 1);
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
	const uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (v >> 32));
	return ((uint64_t) hi << 32) | lo;
}

// Which helper (bit in hmask) holds a package slot for element `c`?  Lane w looks at helper w's slot: one LDS
// access for the whole team.  Returns a lane mask over helper numbers.
__device__ __forceinline__ uint64_t team_find(unsigned char *smem, const SearchArgs &a, uint32_t hmask, uint32_t c, int lane)
{
	uint64_t u = 0;
	if (lane < TEAM_MAX_WPB && ((hmask >> lane) & 1u)) u = team_view(smem, a, (uint32_t) lane).hdr[lc_slot(c, a.tm_lcslots)];
	return __ballot((uint32_t) u == c && (uint32_t) (u >> 32) >= LC_CLAIMED);
}

__device__ __forceinline__ bool dc_lookup(const uint64_t *dc, uint32_t cmask, uint32_t id, uint32_t &od)
{
	uint32_t s = hash_slot(id, cmask);
	for (uint32_t probe = 0; probe < 24; probe++)
	{
		const uint64_t e = dc[s];
		if ((uint32_t) (e >> 32) == id) { od = (uint32_t) e; return true; }
		if (e == DC_EMPTY) return false;
		s = (s + 1) & cmask;
	}
	return false;
}

// insert (id -> od); lanes of the wave may collide on a slot, nobody else touches the table
__device__ __forceinline__ void dc_insert(uint64_t *dc, uint32_t cmask, uint32_t id, uint32_t od)
{
	const uint64_t e = ((uint64_t) id << 32) | od;
	uint32_t s = hash_slot(id, cmask);
	for (uint32_t probe = 0; probe < 64; probe++)
	{
		const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&dc[s]), (unsigned long long) DC_EMPTY, (unsigned long long) e);
		if (old == DC_EMPTY || (uint32_t) (old >> 32) == id) return;
		s = (s + 1) & cmask;
	}
}

// smallest 64-bit value over the wavefront (uniform); ~0 when every lane passes ~0
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
	const uint32_t h = (uint32_t) (v >> 32);
	const uint32_t hmin = wave_min_u32(h);
	const uint32_t lo = (h == hmin) ? (uint32_t) v : 0xFFFFFFFFu;
	const uint32_t lomin = wave_min_u32(lo);
	return ((uint64_t) hmin << 32) | lomin;
}

// The life of a wave that has no query: help walking siblings until none is left.
template <int FUNC, typename SH, int UREG>
__device__ __forceinline__ void team_help(const SearchArgs &a, unsigned char *smem, TeamCtl *ctl, uint32_t wib, uint32_t wpb, int lane)
{
	const TeamView mine = team_view(smem, a, wib);
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	uint32_t *newid   = reinterpret_cast<uint32_t *>(my + a.off_newid);
	float    *newdist = reinterpret_cast<float *>(my + a.off_newdist);
	const uint32_t cmask = a.tm_dccap - 1;
	for (;;)
	{
		// ---- pick the walking sibling with the fewest helpers -------------------------------
		uint32_t target = 0xFFFFFFFFu, best = 0xFFFFFFFFu;
		bool pending = false;
		for (uint32_t m = 0; m < wpb; m++)
		{
			if (m == wib) continue;
			const uint32_t st = __builtin_amdgcn_readfirstlane(ctl[m].state);
			if (st == 0) pending = true;
			if (st != 1) continue;
			const uint32_t cnt = (uint32_t) __builtin_popcount(__builtin_amdgcn_readfirstlane(ctl[m].helpers));
			if (cnt < best) { best = cnt; target = m; }
		}
		if (target == 0xFFFFFFFFu)
		{
			if (!pending) return;                      // nobody will walk again
			__builtin_amdgcn_s_sleep(8);
			continue;
		}
		// ---- my region becomes that walk's package store ---------------------------------------
		for (uint32_t i = lane; i < a.tm_dccap; i += 64) mine.dc[i] = DC_EMPTY;
		for (uint32_t i = lane; i < a.tm_lcslots; i += 64) mine.hdr[i] = (uint64_t) LINK_NONE;
		for (uint32_t i = lane; i < 64u * UREG; i += 64) mine.pub[i] = ~0ull;
		mine.ex[lane] = 0;
		wave_sync();
		// (read BEFORE my bit becomes visible: a job the walking wave posts from now on may name me, and must look new to me;
		// reading it after the atomicOr could swallow a job posted in between — the walking wave would wait for me forever)
		uint32_t last_job = __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[target].jobseq));
		if (lane == 0) __hip_atomic_store(&ctl[wib].done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // nothing delivered to THIS walk yet
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if (lane == 0) atomicOr(&ctl[target].helpers, 1u << wib);
		unsigned char *mreg = smem + (size_t) target * a.wave_bytes;
		const float4 *q4 = reinterpret_cast<const float4 *>(mreg);
		const uint32_t *mhtab = reinterpret_cast<const uint32_t *>(mreg + a.off_hash);
		float qnorm = 0.f;
		if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);
		uint32_t dcount = 0;
		uint64_t last_done = ~0ull;                    // the last element I finished, as a candidate key

		while (__builtin_amdgcn_readfirstlane(ctl[target].state) == 1)
		{
			const uint32_t hm = __builtin_amdgcn_readfirstlane(ctl[target].helpers);
			if (!(hm & (1u << wib))) break;
			const TeamView pubv = team_view(smem, a, (uint32_t) __builtin_ctz(hm));
			const uint32_t myrank = (uint32_t) __builtin_popcount(hm & ((1u << wib) - 1u));
			// a scoring job of the walking wave?  (every helper looks; the job names its slice helpers)
			{
				const uint32_t js = __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[target].jobseq));
				if (js != last_job)
				{
					wave_sync();
					const uint32_t job = __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[target].job));
					wave_sync();
					if ((uint32_t) __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[target].jobseq)) != js) continue;   // a newer job is being written: look again
					last_job = js;
					const uint32_t jm = (job >> 7) & 0xFFu, jn = job & 0x7Fu;
					constexpr uint32_t PASS = 4u * SH::RPG;             // rows of one scoring pass = one slice
					static_assert(PASS <= SLICE_ROWS, "slice area too small for this shape");
					const uint32_t lo = PASS + PASS * (uint32_t) __builtin_popcount(jm & ((1u << wib) - 1u));
					if ((jm & (1u << wib)) && lo < jn)
					{
						const uint32_t cnt = jn - lo < PASS ? jn - lo : PASS;
						const uint32_t *ids = team_view(smem, a, (job >> 15) & 7u).miss + lo;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						score_rows_fit<FUNC, SH::KB, SH::RPG, SLICE_ROWS>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, cnt, ctl[wib].slice, lane);
						wave_sync();
						if (lane == 0) __hip_atomic_store(&ctl[wib].done, js * 8u + target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
					continue;
				}
				if (myrank >= a.tm_spec) { __builtin_amdgcn_s_sleep(1); continue; }     // a slice helper does not speculate
			}
			// snapshot of the main's accepted set: candidate keys (dist, ~idx) of the unexpanded elements
			uint64_t ck[UREG];
			const uint32_t ex = pubv.ex[lane];
#pragma unroll
			for (int k = 0; k < UREG; k++)
			{
				const uint64_t key = pubv.pub[k * 64 + lane];
				const bool open = !((ex >> k) & 1u) && (uint32_t) (key >> 32) != 0xFFFFFFFFu && (uint32_t) key < a.n;
				ck[k] = open ? (key ^ 0xFFFFFFFFull) : ~0ull;
			}
			// the myrank-th smallest that nobody has packaged or claimed yet (at most myrank + 4 steps down the order)
			uint64_t prev = 0, pick = ~0ull;
			bool first = true;
			for (uint32_t step = 0; step < myrank + 5; step++)
			{
				uint64_t m = ~0ull;
#pragma unroll
				for (int k = 0; k < UREG; k++)
				{
					const bool ok = first || ck[k] > prev;
					m = (ok && ck[k] < m) ? ck[k] : m;
				}
				m = wave_min_u64(m);
				if (m == ~0ull) break;
				prev = m; first = false;
				if (step < myrank) continue;
				if (m != last_done && team_find(smem, a, hm, ~(uint32_t) m, lane) == 0) { pick = m; break; }
			}
			if (pick == ~0ull) { __builtin_amdgcn_s_sleep(2); continue; }
			uint32_t cand = ~(uint32_t) pick;
			// Two elements per pick: the chosen one, then — if it beats every other open candidate — its best
			// neighbour: when the main expands `cand` that neighbour is accepted and popped at once (the case no
			// snapshot can predict: 15-17 % of the hops), so its package is prepared right behind cand's.
			uint64_t other = ~0ull;                                        // best open candidate besides the pick
#pragma unroll
			for (int k = 0; k < UREG; k++) other = (ck[k] != pick && ck[k] < other) ? ck[k] : other;
			other = wave_min_u64(other);
			for (int depth = 0; depth < 2; depth++)
			{
				uint64_t hc0 = 0;
				if ((TEAM_COUNT && a.team_dbg)) hc0 = __builtin_amdgcn_s_memtime();
				if (dcount + 2 * a.lstride > a.tm_dccap - a.tm_dccap / 4)   // memo nearly full: start over
				{
					for (uint32_t i = lane; i < a.tm_dccap; i += 64) mine.dc[i] = DC_EMPTY;
					dcount = 0;
					wave_sync();
				}
				const uint32_t slot = lc_slot(cand, a.tm_lcslots);
				if (lane == 0) mine.hdr[slot] = (uint64_t) cand | ((uint64_t) LC_CLAIMED << 32);
				wave_sync();
				uint64_t bestnb = ~0ull;                                    // best scored neighbour, as a candidate key
				for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)                // hnswalg.cpp:76-77 + :95-97, ahead of time
				{
					const uint32_t j = j0 + lane;
					const uint32_t t = a.links[(size_t) cand * a.lstride + (j < a.lstride ? j : a.lstride - 1)];
					bool need = j < a.lstride && t != LINK_NONE && t < a.n;
					if (need && a.hcap) need = !tagset_contains(mhtab, a.hcap / 4u, a.hmagic, t);
					uint32_t od = OD_MISSING;
					bool score = need;
					if (need && dc_lookup(mine.dc, cmask, t, od)) score = false;          // already scored for another element
					const uint64_t sm = __ballot(score);
					const uint32_t ns = (uint32_t) __builtin_popcountll(sm);
					const uint32_t k = lane_rank(sm);
					if (ns)
					{
						if (score) newid[k] = t;
						wave_sync();
						const uint32_t *ids = newid;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, ns, newdist, lane);
						wave_sync();
						const uint32_t od_s = ord_f32(finish_dist<FUNC>(newdist[k & 63], newdist[OUT2 + (k & 63)], qnorm));
						if (score) { od = od_s; dc_insert(mine.dc, cmask, t, od); }
						dcount += ns;
						wave_sync();
					}
					// bit 31 of the id = "the main had visited it when I looked" (ids stay below 2^31 in the beam form):
					// visited is for good, so the main skips its own test for those
					const bool seen = j < a.lstride && t != LINK_NONE && t < a.n && !need;
					if (j < a.lstride) mine.pk[slot * a.lstride + j] = (uint64_t) (seen ? (t | PK_VISITED) : t) | ((uint64_t) od << 32);
					const uint64_t nbk = (need && od != OD_MISSING) ? (((uint64_t) od << 32) | (uint32_t) ~t) : ~0ull;
					bestnb = nbk < bestnb ? nbk : bestnb;
				}
				wave_sync();
				if (lane == 0) mine.hdr[slot] = (uint64_t) cand | ((uint64_t) LC_DONE << 32);
				if ((TEAM_COUNT && a.team_dbg) && lane == 0)
				{
					atomicAdd(a.team_dbg + 11, 1u);
					atomicAdd(a.team_dbg + 12, (uint32_t) (__builtin_amdgcn_s_memtime() - hc0));
				}
				if (depth) break;
				bestnb = wave_min_u64(bestnb);
				if (bestnb >= other) break;                                 // some known candidate is popped first
				const uint32_t nb = ~(uint32_t) bestnb;
				if (__builtin_amdgcn_readfirstlane(ctl[target].state) != 1 || team_find(smem, a, hm, nb, lane)) break;
				cand = nb;
			}
			last_done = pick;
		}
		if (lane == 0) atomicAnd(&ctl[target].helpers, ~(1u << wib));
	}
}

// Cold arguments.  SearchArgs is 88 dwords passed by value; hipcc loads every field a kernel names into scalar registers at its entry and
// keeps them there — the team form held 308-318 spilled SGPRs in round 5 (profiles/r5_hop_budget.md), and about half of the lane exchanges in
// its pop and package look-up sections were reloads of spilled arguments.  Most fields are needed once per QUERY (outputs, the query, the
// ticket) or in the emit step only; those are now read where they are used, with scalar loads from the kernel-argument segment itself (it IS
// memory, and stays valid for the life of the launch), as volatile reads so that the loads are not hoisted back to the entry.  Only what the hop loop reads is named through `a` (and so stays in registers).
#ifdef PGEMB_SIMT_EMULATOR
typedef const SearchArgs *ColdArgs;
__device__ __forceinline__ ColdArgs cold_args(const SearchArgs &a) { return &a; }
#else
typedef const volatile __attribute__((address_space(4))) SearchArgs *ColdArgs;      // (volatile: a read stays where it is written — no hoisting, no merging)
__device__ __forceinline__ ColdArgs cold_args(const SearchArgs &)
{
	return (ColdArgs) __builtin_amdgcn_kernarg_segment_ptr();      // (the kernel's only parameter: offset 0)
}
#endif

template <int FUNC, typename SH, int UREG, bool TEAM = false, bool LEAN = false>
__global__ __launch_bounds__(TEAM ? 512 : 256, (UREG >= 16 && SH::MIN_WAVES > 2) ? 2 : SH::MIN_WAVES) void hnsw_search_kernel_beam(const SearchArgs a)
{
	constexpr uint32_t UCAP = 64u * UREG;
#ifdef HNSW_NO_EARLY_POP
	constexpr bool EARLY_POP = false;
#else
	constexpr bool EARLY_POP = !TEAM;              // banner "Early pop" at the hop loop
#endif
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float        *qf      = reinterpret_cast<float *>(my);
	const float4 *q4      = reinterpret_cast<const float4 *>(my);
	uint32_t     *htab    = reinterpret_cast<uint32_t *>(my + a.off_hash);
	const uint32_t hmask  = a.hcap - 1;
	uint32_t     *newid   = reinterpret_cast<uint32_t *>(my + a.off_newid);
	float        *newdist = reinterpret_cast<float *>(my + a.off_newdist);

	const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wib;
	clock_stamp(cold_args(a)->health, slot, lane, 0);
	uint32_t *vis  = a.vis + (size_t) slot * cold_args(a)->vis_words;
	uint32_t *vlog = a.vlog + (size_t) slot * a.logcap;
	uint64_t *scratch = a.beam_scratch + (size_t) slot * UCAP;
	const uint32_t ef = a.ef;
	bool aborted = false;              // the host asked this launch to end (abort word)
	TeamCtl *ctl = reinterpret_cast<TeamCtl *>(smem + a.off_ctl);
	const uint32_t wpb = blockDim.x >> 6;
	// Stream mode (TEAM kernels; host side: hnsw_gpu_stream_*, hnsw_gpu.hip).  One query per call from hundreds of backends makes
	// small launches, and a query that arrives while every launch in flight is busy waits for a launch to END before it can even
	// start (0.4-0.5 ms of a 1.8 ms round trip at 1 024 backends, profiles/r4f_server_walkers_and_breakdown.txt).  A stream is ONE
	// resident launch: its walking waves take tickets as always, but a ticket is a slot of a ring in pinned host memory that the
	// host fills while the kernel runs — the wave that holds ticket t waits until the host has PUBLISHED more than t queries, walks
	// query t & (ring - 1), writes its results and completion flag into the same slot of the result ring (streamed completion, as
	// the server's launches do), and takes the next ticket.  A query starts the moment a walking wave is free, never later.
	//   * block 0 of the grid is the doorbell (the first block the dispatcher places: it must be resident for anybody to make
	//     progress, so the grid is never larger than what the device holds at once): its first wave copies the host's two control words (published count, stop)
	//     into device memory about once a microsecond; every other wave polls the device copy (L2), so an idle stream costs one
	//     read across the host link per microsecond, not one per resident wave;
	//   * the team geometry is fixed for the life of the stream (team_mains walking waves per block, the others help them: the
	//     helpers of a block have walking siblings for as long as the stream lives, so the helper protocol runs unchanged);
	//   * stop: every wave leaves at its next look — a walking wave after its current query (the host stops a stream when nothing
	//     is outstanding, or gives the stragglers up).  The abort word works as in every launch and also ends the wave.
	const bool stream = TEAM && cold_args(a)->stream_host != nullptr;
	// which optional outputs this launch writes (bit 0 pop sequence, 1 evaluation trace, 2 clock stamps): one word the walk can test
	const uint32_t opt_out = LEAN ? 0u : ((cold_args(a)->out_pops ? 1u : 0u) | (cold_args(a)->out_evals ? 2u : 0u) | (cold_args(a)->out_times ? 4u : 0u));
	if (stream && blockIdx.x == 0)
	{
		if (wib != 0) return;
		const uint32_t *stream_host = cold_args(a)->stream_host;
		uint32_t *stream_dev = cold_args(a)->stream_dev;
		for (;;)
		{
			const uint32_t pub = __hip_atomic_load(stream_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			const uint32_t stop = __hip_atomic_load(stream_host + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");              // what the host wrote before it published is visible to loads issued from here on
			{
				uint32_t *copy = stream_dev + (uint32_t) lane * STREAM_COPY_WORDS;       // (64 lanes = STREAM_COPIES copies)
				__hip_atomic_store(copy, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				if (stop) __hip_atomic_store(copy + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (__builtin_amdgcn_readfirstlane((int) stop)) return;
			__builtin_amdgcn_s_sleep(24);
		}
	}
	if (TEAM)
	{
		if (lane == 0) { ctl[wib].state = wib < a.team_mains ? 0u : 2u; ctl[wib].helpers = 0u; ctl[wib].job = 0u; ctl[wib].jobseq = 0u; ctl[wib].done = 0u; }
		__syncthreads();
	}

	for (;;)
	{
		ColdArgs c = cold_args(a);                                    // this query's cold arguments: loaded here, dead before the walk
		if (TEAM && wib >= c->team_mains) break;                      // this wave only ever helps
		uint32_t qi = 0;
		if (lane == 0) qi = atomicAdd(c->ticket, 1u);
		qi = __builtin_amdgcn_readfirstlane(qi);
		if (stream)
		{
			bool leave = false;
			const uint32_t *ctlw = c->stream_dev + ((blockIdx.x * wpb + wib) % STREAM_COPIES) * STREAM_COPY_WORDS;
			for (uint32_t nap = 8;;)                                      // wait until the host has published query qi (or says stop)
			{
				uint32_t pub = 0, stop = 0;
				if (lane == 0)
				{
					// (relaxed: an acquire here would invalidate this CU's vector cache at every look of every idle wave, under the
					// walking waves' feet — one acquire fence when the wait is over)
					pub = __hip_atomic_load(ctlw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					stop = __hip_atomic_load(ctlw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				pub = __builtin_amdgcn_readfirstlane(pub); stop = __builtin_amdgcn_readfirstlane(stop);
				if ((int32_t) (pub - qi) > 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); break; }   // (wrap-safe: counters mod 2^32)
				if (stop) { leave = true; break; }
				if (nap == 8) __builtin_amdgcn_s_sleep(8); else if (nap == 32) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(127);
				if (nap > 128) __builtin_amdgcn_s_sleep(127);                // (a wave that has waited long looks every ~7 us)
				nap = nap < 512 ? nap * 4 : 512;
			}
			if (__builtin_amdgcn_readfirstlane((int) leave)) break;
			qi &= c->stream_ring - 1u;                                    // from here on qi is the slot: query, outputs and flag of this ticket
		}
		else if (qi >= c->nq) break;
		// (an abort request is sticky for this wave: it takes the remaining tickets without walking and marks every query it does not
		// answer with count 0xFFFFFFFF, so that the caller of an interrupted launch can tell which rows of its outputs are results)
		if (!aborted && (qi & c->abort_mask) == 0u && abort_word_set(c->abort_word)) aborted = true;
		if (__builtin_amdgcn_readfirstlane((int) aborted))
		{
			if (lane == 0) c->out_counts[qi] = ABORTED_COUNT;
			if (stream) break;                                            // (a stream has no last ticket to run to)
			continue;
		}
		if (!LEAN && (opt_out & 4u) && lane == 0) c->out_times[2 * (size_t) qi] = __builtin_amdgcn_s_memrealtime();

		const float *qsrc = c->queries + (size_t) qi * c->q_stride;
		const uint32_t qdim = c->dim, qpad = c->qpad_floats;
		if (stream)
		{
			// the ring lives in pinned host memory and this slot held another query a ring ago: system-scope loads, so that no cache of
			// the device can answer with the old one
			const uint32_t *qw = reinterpret_cast<const uint32_t *>(qsrc);
			for (uint32_t e = lane; e < qpad; e += 64)
			{
				const uint32_t t = __hip_atomic_load(qw + (e < qdim ? e : qdim - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				qf[e] = (e < qdim) ? __uint_as_float(t) : 0.f;
			}
		}
		else
			for (uint32_t e = lane; e < qpad; e += 64)
			{
				const float t = qsrc[e < qdim ? e : qdim - 1];
				qf[e] = (e < qdim) ? t : 0.f;
			}
		wave_sync();
		float qnorm = 0.f;
		if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);
		if (FUNC == F_COSINE_REF) qnorm = query_norm_ref(qf, a.nchunks * 4, lane);

		uint64_t uk[UREG];
#pragma unroll
		for (int k = 0; k < UREG; k++) uk[k] = ~0ull;
		uint32_t ex = 0;                     // bit k: slot k*64+lane has been expanded
		uint32_t usize = 0, logn = 0, evals = 0, hops = 0, hcount = 0;
		uint32_t bstale = 0xFFFFFFFFu;       // ord() of a valid upper bound of the reference's lowerBound
		bool spill = a.hcap == 0;
		uint32_t hs_pop = 0, hs_link = 0, hs_vis = 0, hs_score = 0, hs_acc = 0, hs_q0 = 0;
		uint32_t hc_new = 0, hc_todo = 0, hc_iter = 0, hc_acc = 0, hc_fast = 0, hc_prune = 0, hc_pass2 = 0, hc_acc_loop = 0;   // (diagnostic build: what the accept section does per hop)
		uint32_t jobseq = TEAM ? (uint32_t) __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[wib].jobseq)) : 0u;   // scoring jobs posted by this wave so far
		if (HOP_STAMPS && a.team_dbg) hs_q0 = hop_stamp();
		const uint32_t hnb = a.hcap / 4u;                                  // buckets of the visited set
		if (a.hcap)
		{
			uint4 *h4 = reinterpret_cast<uint4 *>(htab);
			for (uint32_t i = lane; i < a.hcap / 4; i += 64) h4[i] = make_uint4(0u, 0u, 0u, 0u);
			wave_sync();
		}

		if (a.n > 0)
		{
			const uint32_t ep = c->entry;                                  // hnswalg.cpp:55-65
			{
				auto one = [ep](uint32_t) { return ep; };
				score_rows<FUNC, SH::KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, one, 1u, newdist, lane);
			}
			wave_sync();
			const float d0 = finish_dist<FUNC>(newdist[0], newdist[OUT2], qnorm);
			evals = 1;
			if (!LEAN && (opt_out & 2u) && c->evals_cap && lane == 0) c->out_evals[(size_t) qi * c->evals_cap] = ep;
			beam_set<UREG>(uk, 0, ((uint64_t) ord_f32(d0) << 32) | ep, lane);
			usize = 1;
			if (lane == 0)
			{
				if (spill) { vis[ep >> 5] = 1u << (ep & 31); vlog[0] = ep; }
				else { uint32_t eb, et; tagset_split(ep, hnb, a.hmagic, eb, et); htab[4u * eb] = et; }   // empty table: slot 0 of its bucket
				// Nobody helps this walk yet.  A helper of my PREVIOUS walk that was in the middle of a step when that walk ended
				// may not have seen state 0 in between (1 -> 0 -> 1): its region still holds packages scored against the previous
				// query, and its bit would make me read them (and name it in a scoring job).  Without its bit it leaves at its next
				// look (team_help) and attaches again with a clean region.  (Only a wave that takes a second query while siblings
				// help can meet this: a small launch whose other blocks started late.)
				if (TEAM) __hip_atomic_store(&ctl[wib].helpers, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (lane 0 only; helpers set their bits with atomicOr)
			}
			logn = spill ? 1 : 0;
			hcount = 1;
			wave_sync();
			if (TEAM && lane == 0) ctl[wib].state = 1u;                     // helpers may attach from here on

			// Early pop (round 6, one-wave kernels).  The next hop's pop is the best open element of the set as it stands NOW (after this hop's pop)
			// unless a row accepted in this hop beats it — and that scan (per-lane minimum, a 6-step wave minimum, the slot from equality ballots:
			// ~800 cycles of a wave's time) depends on nothing the link-list fetch returns.  So it is taken while that fetch is in flight (nx_*),
			// and the hop's accept step only checks whether an accepted row lies below it (one ballot): if none does, the next hop starts from
			// nx_* at once (~84 % of the hops: what the team form's helpers call the predicted next pop); if one does, or the set was pruned
			// (slots move), the next hop scans as before.  Same pops, same order.
			bool nx_valid = false, nx_has = false;       // nx_*: best open element after this hop's pop, taken during the link-list fetch
			uint32_t nx_slot = 0;
			uint64_t nx_key = 0;
			for (;;)                                                        // hnswalg.cpp:67-112
			{
				// helpers attached to this walk (wave-uniform; read early, used after the pop)
				uint32_t hm = 0;
				if (TEAM) hm = __builtin_amdgcn_readfirstlane(ctl[wib].helpers);
				uint32_t cslot;
				uint64_t ckey;
				uint32_t hs0 = 0, hs1 = 0;
				if (HOP_STAMPS && a.team_dbg) hs0 = hop_stamp();
				uint32_t n_lt = 0;                                          // elements of the set below the popped distance (the stop test's count)
				if (EARLY_POP && nx_valid)
				{
					if (!nx_has) break;                                     // candidateSet empty
					cslot = nx_slot; ckey = nx_key;
					uint32_t cd = (uint32_t) (ckey >> 32);
					cd = (uint32_t) __builtin_amdgcn_readfirstlane((int) cd); // a 32-bit scalar of its own: hipcc otherwise compares (key >> 32) with (ckey >> 32) as 64-bit pairs
					n_lt = beam_count_lt<UREG>(uk, cd);
				}
#ifdef HNSW_OLD_POP_COUNT
				else
				{
					if (!beam_next<UREG>(uk, ex, cslot, ckey)) break;
					uint32_t cd = (uint32_t) (ckey >> 32);
					cd = (uint32_t) __builtin_amdgcn_readfirstlane((int) cd);
					n_lt = beam_count_lt<UREG>(uk, cd);
				}
#else
				else if (!beam_next<UREG, true>(uk, ex, cslot, ckey, &n_lt)) break;     // candidateSet empty
#endif
				nx_valid = false;
				bool nx_taken = false, nx_beaten = false;                   // this hop: the scan was made / an accepted row lies below its result
				if (n_lt >= ef) break;                                      // :70-71  best candidate > lowerBound
				const uint32_t cur = ~(uint32_t) ckey;
				ex |= ((uint32_t) lane == (cslot & 63)) ? (1u << (cslot >> 6)) : 0u;   // :73 pop
				if (!LEAN && (opt_out & 1u))                                // (system scope: a host that polls the sequence sees it as the walk goes)
				{
					ColdArgs cw = cold_args(a);
					const uint32_t pcap = cw->pops_cap;
					uint32_t *pout = cw->out_pops;
					if (hops < pcap && lane == 0)
						__hip_atomic_store(pout + (size_t) qi * pcap + hops, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
				hops++;
				if (!LEAN && (hops & 255u) == 0u && abort_word_set(cold_args(a)->abort_word)) { aborted = true; break; }
				if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_pop += hs1 - hs0; hs0 = hs1; }
				TeamView h0v = {};
				if (TEAM && (TEAM_COUNT && a.team_dbg) && lane == 0) { atomicAdd(a.team_dbg + 5, 1u); if (hm) atomicAdd(a.team_dbg + 0, 1u); }
				uint64_t tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
				if (TEAM && (TEAM_COUNT && a.team_dbg)) tc0 = __builtin_amdgcn_s_memtime();
				if (TEAM && hm)                                             // publish the accepted set for the helpers
				{
					h0v = team_view(smem, a, (uint32_t) __builtin_ctz(hm));
#pragma unroll
					for (int k = 0; k < UREG; k++) h0v.pub[k * 64 + lane] = uk[k];
					h0v.ex[lane] = ex;
				}
				// a helper's package for this element?  (link list + distances of its unvisited neighbours)
				bool have_pk = false;
				const uint64_t *pkp = nullptr;
				const uint64_t *pkh = nullptr;
				if (TEAM && hm)
				{
					const uint64_t who = team_find(smem, a, hm, cur, lane);
					if (who)
					{
						const TeamView pv = team_view(smem, a, (uint32_t) __builtin_ctzll(who));
						const uint32_t sl = lc_slot(cur, a.tm_lcslots);
						pkh = pv.hdr + sl;
						pkp = pv.pk + (size_t) sl * a.lstride;
						uint64_t u = uniform_u64(*pkh);
						uint32_t spins = 0;                                 // in flight: the helper is ahead of any fetch started now
						for (; spins < 4000 && (uint32_t) u == cur && (uint32_t) (u >> 32) == LC_CLAIMED; spins++)
						{
							__builtin_amdgcn_s_sleep(1);
							u = uniform_u64(*pkh);
						}
						have_pk = u == ((uint64_t) cur | ((uint64_t) LC_DONE << 32));
						if (spins == 4000) { uint32_t *hw = cold_args(a)->health; if (hw && lane == 0) atomicAdd(hw + HEALTH_PACKAGE_TIMEOUTS, 1u); }
						if ((TEAM_COUNT && a.team_dbg) && spins && lane == 0) { atomicAdd(a.team_dbg + 6, spins); atomicAdd(a.team_dbg + 10, 1u); }
					}
				}

				for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)               // :76-77
				{
					const uint32_t j = j0 + lane;
					uint32_t t = 0, od_pk = OD_MISSING;
					bool lhit = false;
					if (TEAM && have_pk)
					{
						const uint64_t e = pkp[j < a.lstride ? j : a.lstride - 1];
						wave_sync();                                        // the copy is read before the header is checked again
						lhit = uniform_u64(*pkh) == ((uint64_t) cur | ((uint64_t) LC_DONE << 32));   // not re-claimed meanwhile: the copy is whole
						if (lhit) { t = (uint32_t) e; od_pk = (uint32_t) (e >> 32); }
					}
					// neighbours the helper found visited need no test of their own (hnswalg.cpp:91: `continue`)
					const bool pk_seen = TEAM && lhit && t != LINK_NONE && (t & PK_VISITED);
					if (pk_seen) t = LINK_NONE;
					if (!lhit)
						t = a.links[(size_t) cur * a.lstride + (j < a.lstride ? j : a.lstride - 1)];
					if (EARLY_POP && j0 == 0)                               // while the link list is on its way
					{
						nx_has = beam_next<UREG>(uk, ex, nx_slot, nx_key);
						nx_taken = true;
					}
					if (TEAM && (TEAM_COUNT && a.team_dbg) && lhit && lane == 0) atomicAdd(a.team_dbg + 1, 1u);
					if (TEAM && (TEAM_COUNT && a.team_dbg)) { __builtin_amdgcn_s_waitcnt(0); tc1 = __builtin_amdgcn_s_memtime(); }
					if (HOP_STAMPS && a.team_dbg) { __builtin_amdgcn_s_waitcnt(0); hs1 = hop_stamp(); hs_link += hs1 - hs0; hs0 = hs1; }
					bool isnew = false;
					bool tobits = false;                                    // this id lives in the HBM bitmap (bucket full, or no LDS set)
					if (j < a.lstride && t != LINK_NONE)                    // :91-93
					{
						tobits = spill;
						if (!spill)
						{
							const int r = tagset_test_and_set(htab, hnb, a.hmagic, t);
							isnew = r == TS_NEW;
							tobits = r == TS_FULL;
						}
						if (tobits)
						{
							const uint32_t bit = 1u << (t & 31);
							const uint32_t old = atomicOr(&vis[t >> 5], bit);
							isnew = !(old & bit);
						}
					}
					{
						const uint64_t lm = __ballot(isnew && tobits);      // bits to undo after the query
						if (lm)
						{
							const uint32_t lp = logn + lane_rank(lm);
							if (isnew && tobits && lp < a.logcap) vlog[lp] = t;
							logn += (uint32_t) __builtin_popcountll(lm);
						}
					}
					const uint64_t mask = __ballot(isnew);
					const uint32_t nnew = (uint32_t) __builtin_popcountll(mask);
					if (nnew == 0)
					{
						if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_vis += hs1 - hs0; hs0 = hs1; }
						continue;
					}
					const uint32_t rank = lane_rank(mask);
					uint32_t *ev_out = nullptr;                             // (cold arguments are read in uniform control flow only)
					uint32_t ev_cap = 0;
					if (!LEAN && (opt_out & 2u)) { ColdArgs cw = cold_args(a); ev_out = cw->out_evals; ev_cap = cw->evals_cap; }
					if (isnew)
					{
						newid[rank] = t;
						if (!LEAN && ev_out && evals + rank < ev_cap) ev_out[(size_t) qi * ev_cap + evals + rank] = t;   // (measurement: the rows this walk scores, in order)
						if (TEAM && hm) reinterpret_cast<uint32_t *>(newdist)[rank] = od_pk;      // packaged distance, link order kept
					}
					wave_sync();
					// What still has to be scored here: everything (no helpers), or what no package held
					const uint32_t *sids = newid;
					uint32_t nscore = nnew, krank = (uint32_t) lane, od_c = 0;
					bool hit = false;
					if (TEAM && hm)
					{
						// distances a helper packaged are taken as they are (link order kept by the compaction above)
						const uint32_t myid = newid[lane];
						od_c = reinterpret_cast<const uint32_t *>(newdist)[lane];
						hit = (uint32_t) lane < nnew && od_c != OD_MISSING;
						const uint64_t all = nnew >= 64 ? ~0ull : ((1ull << nnew) - 1ull);
						const uint64_t missm = ~__ballot(hit) & all;
						nscore = (uint32_t) __builtin_popcountll(missm);
						krank = lane_rank(missm);
						if ((TEAM_COUNT && a.team_dbg) && lane == 0)
						{
							atomicAdd(a.team_dbg + 2, nnew); atomicAdd(a.team_dbg + 3, nnew - nscore);
							if (nscore) atomicAdd(a.team_dbg + 4, 1u);
						}
						if (nscore)
						{
							wave_sync();                                    // od_c is in registers before newdist is reused
							if (!hit && (uint32_t) lane < nnew) h0v.miss[krank] = myid;
							wave_sync();
						}
						sids = h0v.miss;
					}
					uint32_t od_mine = od_c;
					if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_vis += hs1 - hs0; hs0 = hs1; }
					if (nscore)
					{
						const uint32_t *ids = sids;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						constexpr uint32_t PASS = 4u * SH::RPG;                 // rows of one scoring pass = one slice
						uint32_t jm = 0;                                        // helpers given slice k = rows PASS * (1 + k) .. (banner at TeamCtl)
						if (TEAM && hm && nscore > PASS)
						{
							uint32_t m = hm;
							for (uint32_t i = 0; i < a.tm_spec && m; i++) m &= m - 1;          // the speculating helpers stay out of it
							const uint32_t want = (nscore - 1) / PASS;
							for (uint32_t i = 0; i < want && m; i++) { jm |= m & (0u - m); m &= m - 1; }
							if (jm)
							{
								++jobseq;
								if (lane == 0)
									__hip_atomic_store(&ctl[wib].job, ((uint32_t) __builtin_ctz(hm) << 15) | (jm << 7) | nscore, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
								wave_sync();                                    // (one wave's LDS operations are served in issue order)
								if (lane == 0)
									__hip_atomic_store(&ctl[wib].jobseq, jobseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							}
						}
						if (jm == 0)
							score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nscore, newdist, lane);
						else
						{
							score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, PASS, newdist, lane);
							const uint32_t covered = PASS + PASS * (uint32_t) __builtin_popcount(jm);   // what I and the slice helpers take; the rest (few helpers) is mine too
							if (covered < nscore)
							{
								auto rest = [ids, covered](uint32_t r) { return ids[covered + r]; };
								score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, rest, nscore - covered, newdist + covered, lane);
							}
							uint32_t lo = PASS, ngot = 0;
							for (uint32_t mm = jm; mm; mm &= mm - 1, lo += PASS)
							{
								const uint32_t h = (uint32_t) __builtin_ctz(mm);
								const uint32_t cnt = nscore - lo < PASS ? nscore - lo : PASS;
								uint32_t polls = 0;
								const uint32_t mine_done = jobseq * 8u + wib;        // this job of THIS walk, nobody else's
								bool got = (uint32_t) __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[h].done)) == mine_done;
								for (; !got && polls < SLICE_WAIT_POLLS; polls++)
								{
									__builtin_amdgcn_s_sleep(0);
									got = (uint32_t) __builtin_amdgcn_readfirstlane(lds_load_u32(&ctl[h].done)) == mine_done;
								}
								wave_sync();
								if (got)
								{
									ngot++;
									if ((uint32_t) lane < cnt)
									{
										newdist[lo + lane] = ctl[h].slice[lane];
										if (FUNC == F_COSINE) newdist[OUT2 + lo + lane] = ctl[h].slice[SLICE_ROWS + lane];
									}
								}
								else                                            // not delivered: the slice is mine (and whatever that helper writes later goes to its own area)
								{
									{ uint32_t *hw = cold_args(a)->health; if (hw && lane == 0) atomicAdd(hw + HEALTH_SLICE_TIMEOUTS, 1u); }
									auto part = [ids, lo](uint32_t r) { return ids[lo + r]; };
									score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, part, cnt, newdist + lo, lane);
								}
							}
							if (ngot) { uint32_t *hw = cold_args(a)->health; if (hw && lane == 0) atomicAdd(hw + HEALTH_SLICES_DELIVERED, ngot); }
						}
						wave_sync();
						const uint32_t od_m = ord_f32(finish_dist<FUNC>(newdist[krank & 63], newdist[OUT2 + (krank & 63)], qnorm));
						od_mine = hit ? od_c : od_m;
					}
					evals += nnew;
					if (TEAM && (TEAM_COUNT && a.team_dbg)) { __builtin_amdgcn_s_waitcnt(0); tc2 = __builtin_amdgcn_s_memtime(); }
					const uint32_t t_mine = newid[lane];
					if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_score += hs1 - hs0; hs0 = hs1; }
					// rows at or above a valid upper bound of lowerBound cannot be accepted (:99)
					uint64_t todo = __ballot((uint32_t) lane < nnew && od_mine < bstale);
					if (HOP_STAMPS && a.team_dbg) { hc_new += nnew; hc_todo += (uint32_t) __builtin_popcountll(todo); if (nscore > 4u * SH::RPG) hc_pass2++; }
					// While the accepted set is below ef every row is accepted whatever its distance (hnswalg.cpp:99: size < ef), one by
					// one in the reference; the set is unordered here, so a hop whose rows ALL fit below ef is appended in one step:
					// row r goes to slot usize + r, fetched by the lane that owns that slot (ds_bpermute), no count, no per-row loop.
					// The first ~ef accepts of every walk — the many-row hops of the descent from the entry point — go this way.
					if (usize + nnew <= ef && (uint32_t) __builtin_popcountll(todo) == nnew)
					{
#pragma unroll
						for (int k = 0; k < UREG; k++)
						{
							const uint32_t lo = (uint32_t) k * 64u;
							if (usize + nnew > lo && usize < lo + 64u)          // (wave-uniform) this register receives rows
							{
								const int r = (int) (lo + (uint32_t) lane) - (int) usize;
								const uint32_t od_r = (uint32_t) __builtin_amdgcn_ds_bpermute((r & 63) << 2, (int) od_mine);
								const uint32_t id_r = (uint32_t) __builtin_amdgcn_ds_bpermute((r & 63) << 2, (int) t_mine);
								const bool take = r >= 0 && (uint32_t) r < nnew;
								uk[k] = take ? (((uint64_t) od_r << 32) | id_r) : uk[k];
							}
						}
						usize += nnew;
						todo = 0;
						if (EARLY_POP) nx_beaten = nx_beaten || !nx_has || __ballot((uint32_t) lane < nnew && (((uint64_t) od_mine << 32) | (uint32_t) ~t_mine) < nx_key) != 0;
						if (HOP_STAMPS && a.team_dbg) { hc_fast++; hc_acc += nnew; }
					}
#ifndef HNSW_SERIAL_ACCEPT
					// Batch form of the accept loop (round 6).  A hop leaves ~4 rows below the stale bound and nearly all of them are accepted
					// (profiles/r6u_*: 3.5 loop iterations, 3.3 accepts per hop), and an accept used to cost a slot write over all UREG
					// registers — 12 of the ~19 VALU instructions of an iteration.  The decisions are still taken one by one in link order,
					// but against the set AS IT STOOD AT THE START OF THE HOP plus the rows accepted so far in this hop:
					//   count_le(set + accepted before r, od_r) = count_le(set, od_r) + |{accepted r' before r : od_r' <= od_r}|
					// — the second term is one ballot over the lanes that hold the new rows — so the set is not touched inside the loop,
					// and the accepted rows are appended in ONE step afterwards: lane r pushes its row to the lane that owns slot
					// usize + (accepted rows before r) (ds_permute; the other lanes push to the lanes behind, so that every lane receives
					// exactly one value).  Same decisions, same set (slot order is not observable).  A hop that could overflow the set
					// (the prune belongs between two accepts) takes the one-by-one loop below.
					if (todo && usize + (uint32_t) __builtin_popcountll(todo) <= UCAP)
					{
						uint64_t acc = 0;                                   // rows of this hop accepted so far (lane mask)
						while (todo)                                        // :99-108, in link order
						{
							const uint32_t r = (uint32_t) __builtin_ctzll(todo);
							todo &= todo - 1;
							if (HOP_STAMPS && a.team_dbg) hc_iter++;
							const uint32_t od = (uint32_t) __builtin_amdgcn_readlane((int) od_mine, (int) r);
							const uint32_t c = beam_count_le<UREG>(uk, od) + (uint32_t) __builtin_popcountll(acc & __ballot(od_mine <= od));
							if (c >= ef)                                    // rejected, and od is a fresh upper bound of lowerBound
							{
								bstale = od < bstale ? od : bstale;
								todo &= __ballot(od_mine < bstale);
								continue;
							}
							acc |= 1ull << r;
						}
						if (acc)
						{
							const uint32_t na = (uint32_t) __builtin_popcountll(acc);
							const uint32_t ra = lane_rank(acc);             // accepted rows in the lanes below mine
							const bool mine_acc = (acc >> lane) & 1ull;
							const uint32_t dst = usize + (mine_acc ? ra : na + ((uint32_t) lane - ra));   // a permutation of the lanes (mod 64)
							const uint32_t od_p = (uint32_t) __builtin_amdgcn_ds_permute((int) ((dst & 63u) << 2), (int) od_mine);
							const uint32_t id_p = (uint32_t) __builtin_amdgcn_ds_permute((int) ((dst & 63u) << 2), (int) t_mine);
							const uint32_t pos = ((uint32_t) lane - usize) & 63u;         // what I received is accepted row number `pos` (if pos < na) ...
							const uint32_t slot = usize + pos;                            // ... and belongs in this slot
							const bool take = pos < na;
							const uint32_t kreg = take ? (slot >> 6) : 0xFFFFFFFFu;       // the register my slot is in (none: I received a row that was not accepted)
#pragma unroll
							for (int k = 0; k < UREG; k++)
								uk[k] = kreg == (uint32_t) k ? (((uint64_t) od_p << 32) | id_p) : uk[k];
							usize += na;
							if (EARLY_POP) nx_beaten = nx_beaten || !nx_has || __ballot(mine_acc && (((uint64_t) od_mine << 32) | (uint32_t) ~t_mine) < nx_key) != 0;
							if (HOP_STAMPS && a.team_dbg) { hc_acc += na; hc_acc_loop += na; }
						}
					}
#endif
					if (EARLY_POP && todo) nx_beaten = true;                // (rare: the one-by-one loop may prune, and then slots move)
					while (todo)                                            // :99-108, in link order, one by one (a hop that may need the prune)
					{
						const uint32_t r = (uint32_t) __builtin_ctzll(todo);
						todo &= todo - 1;
						if (HOP_STAMPS && a.team_dbg) hc_iter++;
						const uint32_t od = (uint32_t) __builtin_amdgcn_readlane((int) od_mine, (int) r);
						if (beam_count_le<UREG>(uk, od) >= ef)              // top().first <= dist and full: rejected,
						{                                                   // and od is a fresh upper bound of lowerBound
							bstale = od < bstale ? od : bstale;
							todo &= __ballot(od_mine < bstale);
							continue;
						}
						const uint32_t t2 = (uint32_t) __builtin_amdgcn_readlane((int) t_mine, (int) r);
						if (usize == UCAP)                                  // make room: drop what lies above the bound
						{
							const uint32_t v = beam_select<UREG>(uk, ef);
							usize = beam_compact<UREG>(uk, ex, v, scratch, lane);
							bstale = v;                                     // lowerBound right now
							if (HOP_STAMPS && a.team_dbg) hc_prune++;
						}
						beam_set<UREG>(uk, usize, ((uint64_t) od << 32) | t2, lane);   // :100,:102
						usize++;
						if (HOP_STAMPS && a.team_dbg) { hc_acc++; hc_acc_loop++; }
					}
					wave_sync();
					if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_acc += hs1 - hs0; hs0 = hs1; }
					if (TEAM && (TEAM_COUNT && a.team_dbg) && lane == 0)
					{
						tc3 = __builtin_amdgcn_s_memtime();
						atomicAdd(a.team_dbg + 7, (uint32_t) (tc1 - tc0));      // pop + stop test + publish + link list
						atomicAdd(a.team_dbg + 8, (uint32_t) (tc2 - tc1));      // visited test + distances (cache or rows)
						atomicAdd(a.team_dbg + 9, (uint32_t) (tc3 - tc2));      // accept loop
					}
				}
				nx_valid = EARLY_POP && nx_taken && !nx_beaten;             // the next hop may start from the early scan
			}
		}

		if (TEAM && lane == 0) ctl[wib].state = 0u;                       // walk over: helpers let go
		c = cold_args(a);                                                  // the emit step's cold arguments
		if (__builtin_amdgcn_readfirstlane((int) aborted))                 // interrupted inside its walk
		{
			if (lane == 0) c->out_counts[qi] = ABORTED_COUNT;
			if (stream) break;
			continue;
		}
		if (!LEAN && (opt_out & 4u) && lane == 0) c->out_times[2 * (size_t) qi + 1] = __builtin_amdgcn_s_memrealtime();
		uint64_t *srt_key = reinterpret_cast<uint64_t *>(my + c->off_res);     // emit scratch (overlays the visited set)
		uint64_t *srt_lab = reinterpret_cast<uint64_t *>(my + c->off_cand);
		uint32_t hs_walk = 0;
		if (HOP_STAMPS && a.team_dbg) hs_walk = hop_stamp();
		// ---- emit: the ef smallest (dist, idx) keys of the set, then the reference's output order ----
		uint32_t rsize = usize;
		if (usize > ef)
		{
			const uint32_t v = beam_select<UREG>(uk, ef);
			rsize = beam_compact<UREG>(uk, ex, v, scratch, lane);          // >= ef, more only with ties at v
		}
#pragma unroll
		for (int k = 0; k < UREG; k++)
		{
			const uint32_t i = (uint32_t) k * 64 + lane;
			if (i < rsize) srt_key[i] = uk[k];
			else if (i < rsize + 3) srt_key[i] = ~0ull;                  // padding of the 4-wide rank loop: never below a key
		}
		wave_sync();
		const uint32_t out_stride = c->out_stride;
		const int mode = c->mode;
		float *out_dists = c->out_dists;
		const size_t obase = (size_t) qi * out_stride;
		const bool sys_out = stream && c->stream_light != 0 && mode == 0;       // (a stream's results: system-scope stores, banner at signal_done)
		uint32_t nout = 0;
		// rank by (dist, idx); only ranks < ef are results (topCandidates, hnswalg.cpp:237-240).  Four broadcast
		// keys per step: the loop is a chain of LDS round trips, not of compares.
		uint32_t myrank[UREG];
#pragma unroll
		for (int k = 0; k < UREG; k++) myrank[k] = 0;
		// (round 6) the labels of the survivors are requested before the ranking instead of behind it (a random 8-byte gather: one memory round
		// trip that the rank loop now covers; at most the ties beyond ef are fetched in vain) ...
		uint64_t lab[UREG];
#pragma unroll
		for (int k = 0; k < UREG; k++)
		{
			lab[k] = 0;
			if (mode != 1 && (uint32_t) k * 64 + lane < rsize) lab[k] = c->labels[(uint32_t) uk[k]];
		}
		// ... and the survivors sit in slots 0 .. rsize - 1, so with rsize <= 32 * UREG (the usual case: rsize = ef, the set's capacity 2 ef) the
		// upper half of the registers is empty and stays out of the loop
#ifndef HNSW_OLD_EMIT
		if (UREG >= 2 && rsize <= 32u * UREG)
		{
			for (uint32_t jx = 0; jx < rsize; jx += 4)
			{
				const uint64_t k0 = srt_key[jx], k1 = srt_key[jx + 1], k2 = srt_key[jx + 2], k3 = srt_key[jx + 3];
#pragma unroll
				for (int k = 0; k < (UREG >= 2 ? UREG / 2 : 1); k++)
					myrank[k] += ((k0 < uk[k]) ? 1u : 0u) + ((k1 < uk[k]) ? 1u : 0u) + ((k2 < uk[k]) ? 1u : 0u) + ((k3 < uk[k]) ? 1u : 0u);
			}
		}
		else
#endif
		for (uint32_t jx = 0; jx < rsize; jx += 4)
		{
			const uint64_t k0 = srt_key[jx], k1 = srt_key[jx + 1], k2 = srt_key[jx + 2], k3 = srt_key[jx + 3];
#pragma unroll
			for (int k = 0; k < UREG; k++)
				myrank[k] += ((k0 < uk[k]) ? 1u : 0u) + ((k1 < uk[k]) ? 1u : 0u) + ((k2 < uk[k]) ? 1u : 0u) + ((k3 < uk[k]) ? 1u : 0u);
		}
#pragma unroll
		for (int k = 0; k < UREG; k++)
			if ((uint32_t) k * 64 + lane >= rsize) myrank[k] = 0xFFFFFFFFu;
		const uint32_t nres = rsize < ef ? rsize : ef;
		if (mode == 1)
		{
#pragma unroll
			for (int k = 0; k < UREG; k++)
				if (myrank[k] < nres)
				{
					c->out_idx[obase + myrank[k]] = (uint32_t) uk[k];
					if (out_dists) out_dists[obase + myrank[k]] = unord_f32((uint32_t) (uk[k] >> 32));
				}
			nout = nres;
			for (uint32_t i = nout + lane; i < out_stride; i += 64)
			{
				c->out_idx[obase + i] = LINK_NONE;
				if (out_dists) out_dists[obase + i] = __builtin_inff();
			}
		}
		else
		{
			// searchKnn, hnswalg.cpp:241-249: labels of the ef results, vacuum filter, (dist, label) order
			uint64_t *out_labels = c->out_labels;
			wave_sync();
			bool tie = false;
#pragma unroll
			for (int k = 0; k < UREG; k++)
			{
				const bool in = myrank[k] < nres;
				if (in)
				{
					srt_key[myrank[k]] = uk[k];                         // sorted by (dist, idx)
					srt_lab[myrank[k]] = lab[k];
				}
			}
			wave_sync();
			// equal distances among the results? (then idx order and label order may differ)
			for (uint32_t i = lane; i + 1 < nres; i += 64)
				if ((uint32_t) (srt_key[i] >> 32) == (uint32_t) (srt_key[i + 1] >> 32)) tie = true;
			const bool any_tie = __ballot(tie) != 0;
			for (uint32_t b = 0; b < nres; b += 64)
			{
				const uint32_t i = b + lane;
				const bool in = i < nres;
				const uint64_t li = in ? srt_lab[i] : 0;
				const uint32_t di = in ? (uint32_t) (srt_key[i] >> 32) : 0;
				const bool keep = in && !((li >> 48) & 1);
				const uint64_t kmask = __ballot(keep);
				uint32_t rank;
				if (!any_tie)
					rank = nout + lane_rank(kmask);
				else
				{
					rank = 0;
					for (uint32_t jx = 0; jx < nres; jx++)
					{
						const uint64_t lj = srt_lab[jx];
						const uint32_t dj = (uint32_t) (srt_key[jx] >> 32);
						const bool kj = !((lj >> 48) & 1);
						rank += (kj && (dj < di || (dj == di && lj < li))) ? 1u : 0u;
					}
				}
				if (keep)
				{
					if (sys_out)
					{
						__hip_atomic_store(out_labels + obase + rank, li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						if (out_dists) __hip_atomic_store(reinterpret_cast<uint32_t *>(out_dists) + obase + rank, __float_as_uint(unord_f32(di)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					}
					else
					{
						out_labels[obase + rank] = li;
						if (out_dists) out_dists[obase + rank] = unord_f32(di);
					}
				}
				nout += (uint32_t) __builtin_popcountll(kmask);
			}
			for (uint32_t i = nout + lane; i < out_stride; i += 64)
			{
				if (sys_out)
				{
					__hip_atomic_store(out_labels + obase + i, (uint64_t) ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					if (out_dists) __hip_atomic_store(reinterpret_cast<uint32_t *>(out_dists) + obase + i, 0x7F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
				else
				{
					out_labels[obase + i] = ~0ull;
					if (out_dists) out_dists[obase + i] = __builtin_inff();
				}
			}
		}
		if (lane == 0)
		{
			if (sys_out) __hip_atomic_store(c->out_counts + qi, nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			else c->out_counts[qi] = nout;
			uint32_t *out_stats = c->out_stats;
			if (out_stats) { out_stats[2 * (size_t) qi] = evals; out_stats[2 * (size_t) qi + 1] = hops; }
		}
		{
			uint32_t *done = c->done;
			if (done) signal_done(done + qi, lane, sys_out);       // (a stream's ring is the library's own coherent pinned memory)
		}

		wave_sync();
		if (logn <= a.logcap)
		{
			for (uint32_t i = lane; i < logn; i += 64) vis[vlog[i] >> 5] = 0u;
		}
		else
		{
			for (uint64_t w = lane, nw = c->vis_words; w < nw; w += 64) vis[w] = 0u;
		}
#ifdef HNSW_ALWAYS_END_WAIT
		if (true)
#else
		if (logn)                                                          // (nothing in the bitmap to wait for: the result stores need no wait, the next query touches none of them)
#endif
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_s_waitcnt(0);
		}
		wave_sync();
		if (HOP_STAMPS && a.team_dbg && lane == 0)
		{
			const uint32_t hs_end = hop_stamp();
			atomicAdd(a.team_dbg + 0, hops);
			atomicAdd(a.team_dbg + 1, hs_pop >> 6);
			atomicAdd(a.team_dbg + 2, hs_link >> 6);
			atomicAdd(a.team_dbg + 3, hs_vis >> 6);
			atomicAdd(a.team_dbg + 4, hs_score >> 6);
			atomicAdd(a.team_dbg + 5, hs_acc >> 6);
			atomicAdd(a.team_dbg + 6, (hs_walk - hs_q0) >> 6);      // query start .. walk over (incl. set-up and entry point)
			atomicAdd(a.team_dbg + 7, (hs_end - hs_walk) >> 6);     // emit + bitmap clean-up
			atomicAdd(a.team_dbg + 8, hc_new); atomicAdd(a.team_dbg + 9, hc_todo); atomicAdd(a.team_dbg + 10, hc_iter); atomicAdd(a.team_dbg + 11, hc_acc);
			atomicAdd(a.team_dbg + 12, hc_fast); atomicAdd(a.team_dbg + 13, hc_prune); atomicAdd(a.team_dbg + 14, hc_pass2); atomicAdd(a.team_dbg + 15, hc_acc_loop);
		}
	}
	if (__builtin_amdgcn_readfirstlane((int) aborted)) { uint32_t *hw = cold_args(a)->health; if (lane == 0) atomicAdd(hw + HEALTH_ABORTED_WAVES, 1u); }
	clock_stamp(cold_args(a)->health, slot, lane, 1);
	if (TEAM)
	{
		if (lane == 0) ctl[wib].state = 2u;
		wave_sync();
		team_help<FUNC, SH, UREG>(a, smem, ctl, wib, wpb, lane);
	}
}

}  // namespace pgemb
