/*
 * hnsw_gpu.h — additive C API of the MI355X (gfx950) HNSW hot path.
 *
 * The reference boundary (hnsw_abi.h == embedding.h:17-56) hands the library ONE
 * query per call and lets it reach the index only through per-element storage
 * callbacks.  One query can never fill an MI355X (≈131 dependent hops), so beside
 * the four drop-in symbols this header adds a batch API over an HBM-resident
 * mirror of the index.  Plain C, opaque handle, int error codes, no exceptions
 * and no framework types cross this line.
 *
 * Each entry point names the reference interface it stands in for.
 */
#ifndef PG_EMBEDDING_AMD_HNSW_GPU_H
#define PG_EMBEDDING_AMD_HNSW_GPU_H

#include "hnsw_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hnsw_gpu_index hnsw_gpu_index;   /* opaque: the device mirror of one index */

enum {
	HNSW_GPU_OK            =  0,
	HNSW_GPU_ERR_HIP       = -1,   /* a HIP runtime call failed (see hnsw_gpu_last_error) */
	HNSW_GPU_ERR_ARG       = -2,   /* bad argument / unsupported configuration           */
	HNSW_GPU_ERR_NOMEM     = -3,
	HNSW_GPU_ERR_INTERNAL  = -4,   /* a device-side invariant tripped                    */
	HNSW_GPU_ERR_NODEVICE  = -5    /* no gfx950 device visible: the library never falls  */
	                               /* back to a CPU path                                 */
};

/* Text of the last failure on the calling thread ("" if none). */
const char *hnsw_gpu_last_error(void);

/* Number of visible HIP devices (0 when there is none). */
int hnsw_gpu_device_count(void);

/* ---------------------------------------------------------------- index mirror */

/* Build the device mirror from `n` element images laid out as the host stores
 * them (embedding.c:222-228: [u32 count][u32 link*maxM][f32*dim][u64 label],
 * meta->size_data_per_element bytes apart).  Stands in for the per-element
 * hnsw_begin_read walk (embedding.c:704-757).  Elements are re-laid-out in HBM as
 * three arrays (links / 16-byte-aligned zero-padded rows / labels). */
int hnsw_gpu_index_create_from_flat(const HnswMetadata *meta, const void *elements, size_t n,
									int device, hnsw_gpu_index **out);

/* An empty mirror with room for `capacity` elements (used by the device builder). */
int hnsw_gpu_index_create_empty(const HnswMetadata *meta, size_t capacity, int device,
								hnsw_gpu_index **out);

/* Append `n` un-linked rows.  vectors: n*dim floats; labels: n values or NULL
 * (label = element number).  The *_dev form takes device pointers and enqueues on
 * `stream` (a hipStream_t, may be NULL).  Equivalent of the "store zero-linked
 * element" half of hnsw_add_point (embedding.c:619-621,670). */
int hnsw_gpu_index_append(hnsw_gpu_index *ix, const coord_t *vectors, const label_t *labels, size_t n);
int hnsw_gpu_index_append_dev(hnsw_gpu_index *ix, const coord_t *d_vectors, const label_t *d_labels,
							  size_t n, void *stream);

/* Link the stored, still un-linked elements [first, first+count) into the graph; every
 * element below `first` must already be linked (or first == 0).  This is hnsw_bind_point
 * (hnswalg.cpp:279-291 = bindPoint :225-232 + mutuallyConnectNewElement :155-223 +
 * getNeighborsByHeuristic :117-153) for many elements.  Elements are processed in batches of
 * min(max_batch, linked/ratio): members of one batch search the graph as it was before the
 * batch.  max_batch = 1 reproduces the reference's serial inserts exactly (graph bytes equal
 * the oracle's); larger batches give a different, equally valid graph much faster.
 * 0 selects the defaults (max_batch 4096, ratio 8).  Enqueued on `stream`, not synchronised. */
int hnsw_gpu_index_link(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio,
						void *stream);

/* Write the mirror back as host element images (inverse of create_from_flat), so a
 * CPU host can search the identical bytes.  `elements` holds count*size_data_per_element. */
int hnsw_gpu_index_export_flat(hnsw_gpu_index *ix, void *elements);

/* Set / clear the vacuum flag of one element's label (embedding.c:920-926). */
int hnsw_gpu_index_set_deleted(hnsw_gpu_index *ix, idx_t idx, int deleted);

/* The same for `count` elements at once (what a VACUUM produces, embedding.c:883-946): one upload of the element
 * numbers and one launch, instead of a blocking copy pair per element. */
int hnsw_gpu_index_set_deleted_batch(hnsw_gpu_index *ix, const idx_t *idx, size_t count, int deleted);

/* Replace / add the elements [first, first+count) from host element images (same layout as
 * create_from_flat; `elements` points at the image of element `first`).  For a host that knows
 * which elements changed (new rows, re-linked neighbours, vacuum flags): incremental mirror
 * maintenance instead of a full re-mirror.  first <= current count. */
int hnsw_gpu_index_update_from_flat(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count);

/* Grow the mirror's capacity (elements and graph are kept). */
int hnsw_gpu_index_reserve(hnsw_gpu_index *ix, size_t capacity);

/* Link list of one element in the host image form: out[0] = count, out[1..maxM] = links
 * (embedding.c:222-226).  `out` holds maxM + 1 values. */
int hnsw_gpu_index_get_links(hnsw_gpu_index *ix, idx_t idx, idx_t *out);

/* The same for element `idx` AND each of its neighbours in one launch: mine = list of idx (maxM + 1 values), others =
 * mine[0] lists of maxM + 1 values each, the j-th being the list of element mine[1 + j] — everything an insert of idx
 * changed (hnswalg.cpp:169-222), for the write-back of hnsw_bind_point.  `others` holds maxM * (maxM + 1) values. */
int hnsw_gpu_index_get_link_lists(hnsw_gpu_index *ix, idx_t idx, idx_t *mine, idx_t *others);

/* hnsw_bind_point's device side in ONE call (hnswalg.cpp:279-291, 225-232): element `idx` — which must be the mirror's current
 * count — is appended (row + label) and linked exactly as the reference's serial insert links it (hnsw_gpu_index_link with
 * max_batch = 1: graph bytes equal the reference's), and the changed link lists come back as hnsw_gpu_index_get_link_lists
 * returns them: `mine` = [count | links] of the new element (maxM + 1 words), `others` = one such list per neighbour in mine's
 * order.  No host wait between the steps: the row is read from and the lists are written to pinned host memory by the kernels
 * themselves, and the calling core polls one completion flag. */
int hnsw_gpu_index_insert_one(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, idx_t *mine, idx_t *others);
/* The same for a caller that has just walked for `point` on this mirror (hnsw_gpu_search_trace in base mode with ef =
 * efConstruction): cand_idx / cand_dist = that walk's result, element numbers and distances ascending by (dist, idx), ncand of them.
 * The insert's own search would return exactly this list (same query, same graph: the new row is not linked yet), so it is not run. */
int hnsw_gpu_index_insert_candidates(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx,
                                     const dist_t *cand_dist, uint32_t ncand, idx_t *mine, idx_t *others);
/* Both run as two launches built for latency (csrc/device_insert.h: every distance getNeighborsByHeuristic can ask for — the lower
 * triangle of the candidates' pair matrix — is scored by all wavefronts at once, the chain of hnswalg.cpp:130-150 then runs out of LDS;
 * one block per reverse link) when the pair matrix of max(efConstruction, maxM + 1) candidates fits a CU's LDS, and through the
 * general builder (hnsw_gpu_index_link with max_batch = 1) otherwise or with HNSW_GPU_INSERT_FUSED=0.  Same graph bytes either way.
 * out[0] / out[1] = inserts of this process that took the first / the second path. */
void hnsw_gpu_insert_path_counts(uint64_t out[2]);

size_t hnsw_gpu_index_count(const hnsw_gpu_index *ix);
/* Elements the mirror has room for (hnsw_gpu_index_reserve grows it: a reallocation and a copy of the whole mirror, so a caller that
 * appends row by row asks for room geometrically and only when this says it must). */
size_t hnsw_gpu_index_capacity(const hnsw_gpu_index *ix);
int    hnsw_gpu_index_device(const hnsw_gpu_index *ix);
void   hnsw_gpu_index_destroy(hnsw_gpu_index *ix);

/* ---------------------------------------------------------------------- search */

/* nq independent hnsw_search() calls (hnswalg.cpp:256-277 = searchKnn(k = ef),
 * hnswalg.cpp:234-252, over searchBaseLayer, hnswalg.cpp:42-114) in one launch.
 *   queries : nq*dim floats
 *   labels  : nq*ef, row q = the reference's result array for query q: ascending
 *             by (distance, label), vacuumed labels removed; unused tail = ~0
 *   dists   : nq*ef distances in the same order (may be NULL; the reference does
 *             not return them, embedding.c:345-351 wishes it did); tail = +inf
 *   counts  : nq result counts (<= ef)
 * Host-pointer form: copies in, runs, copies out, synchronises. */
int hnsw_gpu_search_batch(hnsw_gpu_index *ix, const coord_t *queries, size_t nq, size_t ef,
						  label_t *labels, dist_t *dists, uint32_t *counts);

/* Device-pointer form: everything resident in HBM, enqueued on `stream`
 * (hipStream_t or NULL), no synchronisation.
 *   d_stats : NULL or nq*2 u32 = {distance evaluations E_q, hops H_q} per query
 *             (the counts of hnswalg.cpp:59,96 and :76 — SURVEY.md §8d). */
int hnsw_gpu_search_batch_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
							  label_t *d_labels, dist_t *d_dists, uint32_t *d_counts,
							  uint32_t *d_stats, void *stream);

/* searchBaseLayer() alone (hnswalg.cpp:42-114): element numbers + distances
 * ascending by (distance, idx), no label lookup / vacuum filter.  Device pointers. */
int hnsw_gpu_search_base_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
							 idx_t *d_idx, dist_t *d_dists, uint32_t *d_counts,
							 uint32_t *d_stats, void *stream);

/* Milliseconds the most recent search kernel of this index spent on the device,
 * from HIP events recorded on its stream around the launch (waits for it). */
int hnsw_gpu_last_search_ms(hnsw_gpu_index *ix, float *ms);
/* Same for the launch `back` launches ago (0 = most recent; the last 64 are kept). */
int hnsw_gpu_search_ms(hnsw_gpu_index *ix, unsigned back, float *ms);

/* Where the device time of the last hnsw_gpu_search_batch call (host pointers, more than 16 queries: the copy path) went:
 * out[0] = upload of the queries, out[1] = the search kernel, out[2] = download of labels / distances / counts, in milliseconds
 * (HIP events on the default stream around the three steps; SURVEY.md 8(d): the PCIe share is reported, never hidden). */
int hnsw_gpu_last_batch_ms(hnsw_gpu_index *ix, float out[3]);

/* Symbol of the kernel the mirror's last search launch ran, spelled as rocprofv3 prints it
 * (e.g. "pgemb::hnsw_search_kernel_beam<0, pgemb::Shape12x2, 4>"), so a bench line and a kernel trace
 * can be matched by name. */
int hnsw_gpu_last_search_kernel(hnsw_gpu_index *ix, char *buf, size_t len);

/* Health of the mirror's default search workspace (8 words).  out8[0] = 1 while an abort request is pending, [1] = slices a
 * team helper did not deliver in time and [2] = helper packages that stayed "claimed" past the bound (both were then
 * computed by the walking wave itself: results are unaffected, the counts say a protocol is slower than designed;
 * 0 in a healthy life), [3] = waves that left a launch because of an abort request, [4] = slices team helpers scored
 * for walking waves (says that mechanism is in use), [5] = abort requests this workspace has received, [6..7] = 0 (totals since the mirror exists). */
int hnsw_gpu_index_health(hnsw_gpu_index *ix, uint32_t *out8);

/* Ask the search launches in flight to end: every wave looks at its workspace's abort word at the top of a query
 * and (all kernels but the lean narrow-row one, whose walks are a fraction of a millisecond) every 256 hops of a walk; from
 * then on it takes the remaining queries of the launch without walking.  No wait in the kernels is unbounded, so this is for
 * the unknown: a launch that never ends costs its caller's patience, not the device.  What an aborted launch leaves behind is
 * DEFINED per query: a query it finished has its results and its count, every other query has count HNSW_GPU_COUNT_ABORTED
 * (and no completion flag); the host-pointer calls return an error for such a launch, a caller of the asynchronous
 * device-pointer forms looks at the counts (or at hnsw_gpu_index_health [5] before and after).  The workspace is re-zeroed
 * by the next launch.  Callable from ANY thread, also while another thread is blocked inside a search call on the same mirror.
 * hnsw_gpu_index_abort: the mirror's default workspace only.  hnsw_gpu_abort_all: every workspace of every mirror, context
 * and shard of this process (a process-wide emergency stop; the library itself never uses it); returns how many it reached.
 * HNSW_GPU_WATCHDOG_S=<seconds> in the environment makes the library abort, by itself, the ONE workspace whose launch has
 * been ON the device longer than that (a helper thread, off by default; time spent queued behind other launches does not
 * count); the polled host-pointer calls do the same for the workspace they wait on after HNSW_GPU_POLL_LIMIT_S (default 120). */
#define HNSW_GPU_COUNT_ABORTED 0xFFFFFFFFu
int hnsw_gpu_index_abort(hnsw_gpu_index *ix);
int hnsw_gpu_abort_all(void);

/* Configuration.  Every knob of the library is an optional integer in one table that is filled ONCE, at the library's first
 * use, from the environment (the operational knobs of INTEGRATION.md's table) — no entry point reads the environment on a call
 * path.  A host that wants another value later says so: hnsw_gpu_config_set(name, "value") / (name, NULL) = back to the default;
 * hnsw_gpu_config_get returns 0 and the value, or 1 when the knob is at its default; hnsw_gpu_config_reload re-reads the
 * environment (and resets the knobs that are not environment knobs).  Names are the HNSW_GPU_* names of INTEGRATION.md; test knobs
 * (kernel forms and shapes the host code never picks by itself) can only be set through this call; unknown names are an error. */
int  hnsw_gpu_config_set(const char *name, const char *value);
int  hnsw_gpu_config_get(const char *name, long long *value);
void hnsw_gpu_config_reload(void);

/* Resident query slots (waves) the last search launch used — occupancy figure. */
int hnsw_gpu_last_search_slots(hnsw_gpu_index *ix, uint32_t *slots);

/* Search contexts.  A mirror serialises its own launches (one visited-set workspace); a context
 * is an extra workspace bound to the same mirror, so batches launched through different contexts
 * on different streams run concurrently: while one launch drains (its last, longest queries) the
 * next one already fills the freed CUs.  The index must not be modified while contexts search it. */
typedef struct hnsw_gpu_ctx hnsw_gpu_ctx;
int  hnsw_gpu_ctx_create(hnsw_gpu_index *ix, hnsw_gpu_ctx **out);
void hnsw_gpu_ctx_destroy(hnsw_gpu_ctx *ctx);
/* Walking waves per block for the context's SMALL launches (fewer queries than resident waves run as teams: by default the queries
 * are spread over as many blocks as the device holds and the other waves of a block help — one walk per 8-wave block for a launch of
 * up to one query per CU: the shape of a lone launch).  A host that keeps several small launches in flight at once (the batching
 * server) knows what a single launch cannot: together they ask for more waves than the device has, and every helper then displaces
 * somebody's walk.  per_block = 1..8 makes at least that many waves of a block walk (8 = every wave walks; helpers appear only when
 * the launch's queries run out); 0 = back to the default.  Results do not depend on it. */
int  hnsw_gpu_ctx_set_walkers(hnsw_gpu_ctx *ctx, unsigned per_block);
/* 8-wave team blocks `device` holds at once (one per compute unit for rows wider than 320 floats): with W walks in flight over all
 * launches, ceil(W / blocks) walking waves per block keep every walk on the device.  <= 0: no such device. */
int  hnsw_gpu_device_blocks(int device);
int  hnsw_gpu_search_batch_ctx(hnsw_gpu_ctx *ctx, const coord_t *d_queries, size_t nq, size_t ef,
							   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
							   void *stream);
int  hnsw_gpu_ctx_search_ms(hnsw_gpu_ctx *ctx, unsigned back, float *ms);
/* Host-pointer form of a context search (same arrays as hnsw_gpu_search_batch): copies and launch
 * go to a stream the context owns and only that stream is waited for, so several host threads —
 * one context each — keep several batches in flight (the batching server, hnsw_gpu_server.h).
 * One caller at a time per context. */
int  hnsw_gpu_search_batch_ctx_host(hnsw_gpu_ctx *ctx, const coord_t *queries, size_t nq, size_t ef,
									label_t *labels, dist_t *dists, uint32_t *counts);
/* Streamed completion.  Device-pointer launch on a stream the context owns, plus nq completion
 * flags: when query i's outputs are complete the kernel makes them visible system-wide and then stores
 * 1 to d_done[i].  With d_done and the output arrays in pinned host memory (hnsw_gpu_host_alloc: the
 * device writes it directly) a host thread can hand out each answer when that query's own walk ends
 * instead of when the slowest query of the batch does — a walk is ~160 dependent hops and the slowest
 * of a batch takes 2-3x the mean.  The queries may live in pinned host memory too.  The caller zeroes
 * d_done before the call; nothing is waited for.  hnsw_gpu_ctx_idle: 1 once the context's last launch
 * has left the device, 0 while it runs, < 0 on error. */
int  hnsw_gpu_search_batch_ctx_flags(hnsw_gpu_ctx *ctx, const coord_t *d_queries, size_t nq, size_t ef,
									 label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
									 uint32_t *d_done);
int  hnsw_gpu_ctx_idle(hnsw_gpu_ctx *ctx);

/* Streams: ONE resident search launch that the host feeds while it runs (csrc/device_search.h, "Stream mode") — the shape for
 * queries that arrive one at a time from many callers (the reference hands the path one query per hnsw_search call, embedding.c:317,
 * from one process per connection): a query starts the moment a walking wave is free instead of waiting for a launch in flight to end.
 *   hnsw_gpu_stream_open    launches the kernel on the context's own stream: every block the device holds, `walkers` (1..8, 0 = 4)
 *                           walking waves per 8-wave block, the others help them; ring = slots of the query / result ring (a power of
 *                           two in [64, 2^20]) in pinned host memory the library allocates; the context serves the stream until
 *                           _close (no other call on it).  Needs the team form of the beam kernel (ef <= 256; <= 512 on wide rows).
 *   hnsw_gpu_stream_buffers the ring: queries ring x dim, labels / dists ring x ef, counts, completion flags ring (host pointers).
 *                           Query number t (t = 0, 1, 2, ... in publication order) lives in slot t & (ring - 1): the host writes its
 *                           dim floats, zeroes the slot's flag, and only then publishes; it reuses a slot when the query that held
 *                           it a ring ago has been answered.
 *   hnsw_gpu_stream_publish published_total = how many queries have been written so far (a counter mod 2^32): a release store — the
 *                           kernel's doorbell wave forwards it to the waiting waves within about two microseconds.
 *   results                 flags[slot] becomes 1 once labels / dists / counts of that slot are complete; rows are what
 *                           hnsw_gpu_search_batch returns.  A stream writes its results with system-scope (write-through) stores and
 *                           stores the flag once they are acknowledged — the L2 write-back of a full release per answered query was
 *                           the ceiling of a many-backend server; HNSW_GPU_STREAM_LIGHT=0 restores plain stores + the full release
 *                           (what hnsw_gpu_search_batch_ctx_flags does for caller-provided buffers).
 *   hnsw_gpu_stream_alive   1 while the launch is on the device.
 *   hnsw_gpu_stream_close   stop: every wave leaves at its next look (a walking wave after its query); waits for the launch to end
 *                           (a launch that does not end within 2 s is asked through its abort word), frees the ring.  Queries
 *                           published but not yet started are dropped: close a stream when nothing is outstanding.
 *   hnsw_gpu_stream_abandon the same stop, but neither the ring nor the handle is freed (leaked on purpose): for a host that had to give
 *                           a stream up while it could not prove that none of its own threads was still inside the ring; such a late
 *                           thread may still call _publish / _buffers on the handle (they touch the leaked memory only), nothing else.
 * Results do not depend on the mode: a query's walk is the same walk in a plain launch, a team launch or a stream. */
typedef struct hnsw_gpu_stream hnsw_gpu_stream;
int  hnsw_gpu_stream_open(hnsw_gpu_ctx *ctx, size_t ef, size_t ring, unsigned walkers, hnsw_gpu_stream **out);
int  hnsw_gpu_stream_buffers(hnsw_gpu_stream *s, coord_t **queries, label_t **labels, dist_t **dists, uint32_t **counts, uint32_t **flags);
int  hnsw_gpu_stream_publish(hnsw_gpu_stream *s, uint32_t published_total);
int  hnsw_gpu_stream_alive(hnsw_gpu_stream *s);
int  hnsw_gpu_stream_close(hnsw_gpu_stream *s);
int  hnsw_gpu_stream_abandon(hnsw_gpu_stream *s);

/* One query together with its walk: the results of hnsw_gpu_search_batch plus the sequence of elements the walk expanded
 * (candidateSet pops, hnswalg.cpp:73; *npops of them, the first min(*npops, pops_cap) stored) and the number of distance
 * evaluations.  A walk is a deterministic function of the elements it touched — the expanded ones and their link
 * targets — so a caller that finds those byte-identical in another copy of the index (the host's pages) knows that the
 * reference's own walk over that copy returns the same answer: the basis of the drop-in library's validated mirror cache
 * (embedding_shim.cpp).  base != 0: searchBaseLayer only, `labels` receives element numbers (hnsw_gpu_search_base_dev's
 * output widened to 64 bits).  Host pointers. */
int  hnsw_gpu_search_trace(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, label_t *labels, dist_t *dists,
						   uint32_t *count, idx_t *pops, size_t pops_cap, uint32_t *npops, uint32_t *nevals);
/* The same in three steps, for a caller that checks the walk WHILE it runs (the kernel stores every pop into pinned host
 * memory as it happens): _begin launches; _poll copies out the pops that have become visible since the last poll, in
 * order (*finished = the walk is over and every stored pop has been handed out); _end waits and returns the results
 * (*npops = pops of the whole walk, of which at most pops_cap were stored).  One trace at a time per mirror, all three
 * calls from one thread, nothing else on that mirror in between.  No lock is held between the calls; a trace that is
 * never ended (the caller was thrown out of its own code) is waited for by the next _begin. */
int  hnsw_gpu_search_trace_begin(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, size_t pops_cap);
int  hnsw_gpu_search_trace_poll(hnsw_gpu_index *ix, idx_t *pops, size_t max, size_t *got, int *finished);
int  hnsw_gpu_search_trace_end(hnsw_gpu_index *ix, label_t *labels, dist_t *dists, uint32_t *count, uint32_t *npops, uint32_t *nevals);
/* Pinned host memory for the host-pointer entry points (NULL on failure). */
void *hnsw_gpu_host_alloc(size_t bytes);
void  hnsw_gpu_host_free(void *p);

/* ------------------------------------------------------------------- distances */

/* out[i] = hnsw_dist_func(func, q, rows + i*dim, dim)  (distfunc.c:171-174) for
 * i < nrows, one launch.  Host pointers. */
int hnsw_gpu_dist_batch(dist_func_t func, const coord_t *q, const coord_t *rows, size_t nrows,
						size_t dim, dist_t *out);

/* Device form: rows are row_stride floats apart (row_stride % 4 == 0, rows 16-byte
 * aligned, padding zero); q holds dim floats. */
int hnsw_gpu_dist_batch_dev(dist_func_t func, const coord_t *d_q, const coord_t *d_rows,
							size_t nrows, size_t dim, size_t row_stride, dist_t *d_out, void *stream);

/* Exact k nearest rows of the mirror by exhaustive scoring with the same distance
 * code (ground truth for recall; ties broken by lower idx).  Device pointers;
 * d_idx: nq*k. */
int hnsw_gpu_bruteforce_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k,
							idx_t *d_idx, dist_t *d_dists, void *stream);

/* Same result (exact, same distances), but the Q x N scoring runs as a dense f32 contraction on
 * the matrix cores (v_mfma_f32_32x32x2_f32) used as a filter against a per-query bound; the few
 * survivors are re-scored with the canonical distance code (device_bf_mfma.h).  This is the
 * "batched queries as an MFMA GEMM" form of BASELINE config 5; L2 and cosine only (Manhattan
 * falls back to the scan above).  Synchronises `stream`. */
int hnsw_gpu_bruteforce_mfma_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k,
								 idx_t *d_idx, dist_t *d_dists, void *stream);
/* Device milliseconds of the GEMM/filter kernel of the last call above (MFMA roofline figure:
 * 2*nq*n*stride flops). */
float hnsw_gpu_last_bruteforce_gemm_ms(void);
/* queries (= rows) per block tile of that launch: 128 or 256 (256 x 256 tiles are picked for launches with thousands of tiles) */
int hnsw_gpu_last_bruteforce_tile(void);

/* ----------------------------------------------------------------- multi-shard */

/* Merge `nlists` per-shard result lists per query (each ef entries: ascending by
 * (dist, label), tail padded with dist=+inf / label=~0) into the ef best overall.
 * Stands in for nothing in the single-process reference; it is the one exchange
 * step of a row-sharded index.  Layout: d_in_*[list][query][ef]. */
int hnsw_gpu_merge_topk_dev(int device, const label_t *d_in_labels, const dist_t *d_in_dists,
							size_t nlists, size_t nq, size_t ef,
							label_t *d_out_labels, dist_t *d_out_dists, uint32_t *d_out_counts,
							void *stream);

/* Same with the per-list arrays anywhere in memory: list l's labels start at d_in_labels +
 * l*label_list_stride (in label_t), its distances at d_in_dists + l*dist_list_stride (in dist_t).  Lets ONE
 * gathered buffer of per-rank blocks [labels | dists] be merged in place (one all-gather per search). */
int hnsw_gpu_merge_topk_strided_dev(int device, const label_t *d_in_labels, size_t label_list_stride,
									const dist_t *d_in_dists, size_t dist_list_stride, size_t nlists,
									size_t nq, size_t ef, label_t *d_out_labels, dist_t *d_out_dists,
									uint32_t *d_out_counts, void *stream);

/* Wait until everything enqueued so far on `stream` (a hipStream_t; NULL = the device's default stream) of `device` has completed:
 * the one synchronisation a C host without HIP needs around the asynchronous *_dev entry points (hnsw_gpu_server's row-sharded
 * front waits for its merge with it). */
int hnsw_gpu_device_wait(int device, void *stream);

/* A device buffer shared between PROCESSES — the exchange buffer of a row-sharded search whose shards live in different
 * processes (one GPU-owning server per GPU; SURVEY.md §8e "direct P2P stores into rank-0 memory").  The merging process
 * allocates it and hands the 64-byte handle to the others over whatever channel they share; they map it
 * (hnsw_gpu_shared_open: on another GPU their stores into it cross xGMI as peer stores), run hnsw_gpu_search_batch_dev
 * on their shard with the output pointers inside it — list r = the r-th [nq][ef] block — and tell the owner when their
 * stream has drained; the owner merges with hnsw_gpu_merge_topk[_strided]_dev.  No staging copy, no collective library.
 * (HSA_ENABLE_IPC_MODE_LEGACY=0 where the host driver only supports dmabuf IPC.) */
typedef struct hnsw_gpu_ipc_handle { unsigned char bytes[64]; } hnsw_gpu_ipc_handle;
int hnsw_gpu_shared_alloc(int device, size_t bytes, void **d_ptr, hnsw_gpu_ipc_handle *handle);     /* owner */
int hnsw_gpu_shared_open(int device, const hnsw_gpu_ipc_handle *handle, void **d_ptr);            /* another process */
int hnsw_gpu_shared_close(int device, void *d_ptr);                                                /* that process, when done */
int hnsw_gpu_shared_free(int device, void *d_ptr);                                                 /* owner */

/* A row-sharded index inside ONE process (an index larger than one GPU behind a C host; the reference has no
 * counterpart: embedding.c:982 amcanparallel = false).  `shards` are mirrors the caller built — one graph per
 * contiguous row range, labels globally unique (SURVEY.md §8e mode 2) — on one or several devices; they are
 * borrowed, not owned.  A search runs hnsw_search() semantics (hnswalg.cpp:256-277) for every query on every
 * shard concurrently (one stream per shard; a shard on another device gets its own copy of the queries and,
 * with peer access, stores its result lists straight into the merge device's memory over xGMI), then one
 * merge kernel keeps the ef best by (distance, label).  Result = "per-shard search + merge" exactly; parity
 * is defined against the oracle per shard + a CPU merge.  Queries arrive and results leave on the device of
 * shard 0; the *_dev form is enqueued on `stream` of that device and not synchronised. */
typedef struct hnsw_gpu_sharded hnsw_gpu_sharded;
int    hnsw_gpu_sharded_create(hnsw_gpu_index *const *shards, size_t nshards, hnsw_gpu_sharded **out);
void   hnsw_gpu_sharded_destroy(hnsw_gpu_sharded *s);
size_t hnsw_gpu_sharded_nshards(const hnsw_gpu_sharded *s);
int    hnsw_gpu_sharded_search_dev(hnsw_gpu_sharded *s, const coord_t *d_queries, size_t nq, size_t ef,
								   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, void *stream);
int    hnsw_gpu_sharded_search(hnsw_gpu_sharded *s, const coord_t *queries, size_t nq, size_t ef,
							   label_t *labels, dist_t *dists, uint32_t *counts);
/* Where the time of the last sharded call went, from HIP events on each shard's own device (waits for the call): search_ms[i] =
 * shard i's search kernel (with peer access its result stores over xGMI are part of it), peer_ms[i] = what followed on that
 * shard's stream until its lists were in the merge device's buffer (the staged peer copy; ~0 with direct stores), *merge_ms = the
 * merge kernel, direct[i] (may be NULL) = 1 when shard i writes the merge device's memory itself.  Arrays of nshards values. */
int    hnsw_gpu_sharded_last_ms(hnsw_gpu_sharded *s, float *search_ms, float *peer_ms, float *merge_ms, int *direct);

/* Measurement and diagnostic entry points (evaluation traces, replay / gather roofs, team counters, clocks, placement) are declared
 * in hnsw_gpu_diag.h: bench and trace tooling, not part of what a host links against. */

#ifdef __cplusplus
}
#endif
#endif /* PG_EMBEDDING_AMD_HNSW_GPU_H */
