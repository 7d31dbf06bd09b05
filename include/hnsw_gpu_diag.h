/*
 * hnsw_gpu_diag.h — measurement and diagnostics of libhnsw_gpu.so.
 *
 * Nothing here is needed to USE the library (include/hnsw_gpu.h is the product's ABI); these entry points exist for
 * bench.py, the profiling scripts and the tests that price the search kernel against its own trace: evaluation traces,
 * the replay / gather roofs, counters of the team form, the shader clock a launch ran at and where the mirror's arrays
 * sit in the device's address space.  Same library, same calling conventions, no stability promise.
 */
#ifndef PG_EMBEDDING_AMD_HNSW_GPU_DIAG_H
#define PG_EMBEDDING_AMD_HNSW_GPU_DIAG_H

#include "hnsw_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics of the team form of the search kernel (launches with fewer queries than resident waves: idle
 * waves of a block pre-fetch link lists and distances for a sibling's walk).  With HNSW_GPU_TEAM_COUNTERS=1 in
 * the environment the last launch of the mirror's default workspace counted, over all its queries:
 * out[0] hops that had helpers, [1] link lists served from a helper's cache, [2] neighbour ids looked up,
 * [3] distances served from a cache, [4] hops that still scored rows themselves, [5] all hops, [6] polls spent
 * waiting for a helper that had the element in flight, [10] hops that waited, [7]/[8]/[9] shader cycles of the
 * walking waves in: pop + stop test + link list / visited test + distances / accept loop, [11] elements the
 * helpers finished, [12] helper cycles spent on them.  `out` holds 16 values. */
int hnsw_gpu_team_counters(hnsw_gpu_index *ix, uint32_t *out16);

/* Measurement: the same launch as hnsw_gpu_search_batch_dev that also writes its EVALUATION TRACE — d_evals[i * evals_cap + j] =
 * the j-th row query i scored (j < d_stats[2 * i]; truncated at evals_cap), d_times[2 * i], [2 * i + 1] = the device's
 * constant-rate clock (100 MHz) at the start of query i and at the end of its walk (d_times may be NULL) — and the REPLAY ROOF
 * made from it: the rows of such a trace gathered again by `slots` resident waves (hnsw_gpu_last_search_slots of the traced
 * launch, or more) in the same query order, with the search kernel's load shape <kb, rpg> (device_dist.h, score_rows: kb
 * chunk-steps of rpg rows per 16-lane group = kb * rpg 16-byte loads in flight per lane; the search kernel's own shape is <2,2> /
 * <2,4> up to 128 dims, <4,2> up to 256, <8,2> up to 512, <12,2> beyond) and nothing in between.  *ms = best of three repetitions,
 * *bytes = row bytes one repetition reads; word_sum (NULL, or for tests): the sum mod 2^64 of the 32-bit patterns of every word
 * one repetition read for the trace — equal to the same sum over the traced rows of the table.  The search kernel should not beat
 * the best replay of its own trace: search time / replay time is the cost of the walk's dependent chain, bytes / replay time
 * what the memory system gives this access pattern (bench.py: roofline.replay). */
int hnsw_gpu_search_traced_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
                               label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
                               idx_t *d_evals, size_t evals_cap, uint64_t *d_times, void *stream);
int hnsw_gpu_replay_roof(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
                         unsigned slots, int kb, int rpg, float *ms, double *bytes, uint64_t *word_sum);
/* ... with every query's trace cut into `parts` (1..64) equal pieces that different waves gather: the roof of a launch of fewer
 * queries than resident waves in which `parts` waves share the rows of one walk (parts = 1: the call above). */
int hnsw_gpu_replay_roof_parts(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
                               unsigned slots, int kb, int rpg, unsigned parts, float *ms, double *bytes, uint64_t *word_sum);

/* Shader clock (MHz) the most recent search launch of the mirror's default workspace ran at: shader-clock ticks over ticks of
 * the constant 100 MHz clock, both read by the launch's first wave when it starts and when it leaves (waits for the launch).
 * The narrow-row kernel is bound by instruction issue, so its time scales with this clock; the wide-row kernels are bound by
 * HBM and do not care.  *mhz = 0 when the launch recorded nothing (a stream, or a kernel form without the stamps). */
int hnsw_gpu_last_search_clock_mhz(hnsw_gpu_index *ix, double *mhz);

/* Where the mirror and its default search workspace sit in the device's address space: out[2*i] = device address, out[2*i+1] =
 * bytes, for i = 0 arena (one allocation holding rows | links | labels, each on a 2 MiB boundary), 1 rows, 2 links, 3 labels,
 * 4 visited bitmaps, 5 bitmap logs, 6 prune scratch of the beam form, 7 ticket word.  `out` holds 16 values. */
int hnsw_gpu_index_placement(hnsw_gpu_index *ix, uint64_t *out16);

/* Shader clock (MHz) a block of the MFMA filter kernel of the last hnsw_gpu_bruteforce_mfma_dev call saw over its K loop:
 * shader-clock ticks / constant-clock ticks. */
double hnsw_gpu_last_bruteforce_clock_mhz(void);

/* Practical roof of the search kernel's memory access pattern on THIS mirror's row table: independent
 * waves gathering random whole rows with 16-byte loads, `loads_per_lane` (4/8/12/16/24) in flight per lane,
 * `waves_per_cu` resident waves per CU, `iters` gathers per wave; best of three timed repetitions in GB/s.
 * No query of the fused kernel can read rows faster from HBM than this dependency-free gather. */
int hnsw_gpu_gather_roof(hnsw_gpu_index *ix, int loads_per_lane, int waves_per_cu, unsigned iters, float *gbps);

#ifdef __cplusplus
}
#endif
#endif /* PG_EMBEDDING_AMD_HNSW_GPU_DIAG_H */
