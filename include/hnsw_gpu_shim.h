/*
 * hnsw_gpu_shim.h — the few additive calls of libembedding_gpu.so, the library that
 * exports the reference's own four symbols (hnsw_abi.h) on top of libhnsw_gpu.so.
 *
 * They exist because HnswMetadata (embedding.h:28-42) identifies no relation and is
 * re-created for every scan (embedding.c:254): a host that can say "this is the same,
 * unchanged index" (attach) saves the drop-in hnsw_search() the validation reads of
 * its cache (below; round 1 had no cache and re-mirrored the index on every call).
 * INTEGRATION.md shows where the Postgres glue would call them.
 */
#ifndef PG_EMBEDDING_AMD_HNSW_GPU_SHIM_H
#define PG_EMBEDDING_AMD_HNSW_GPU_SHIM_H

#include "hnsw_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Walk the host index through hnsw_begin_read/hnsw_end_read (embedding.c:704-767)
 * and build its device mirror.  The caller owns *out (hnsw_gpu_index_destroy). */
int hnsw_gpu_shim_snapshot(HnswMetadata *meta, hnsw_gpu_index **out);

/* From now on hnsw_search(meta, ...) uses `ix` instead of re-mirroring.  The host
 * promises the index does not change while attached (or re-attaches a fresh mirror). */
int hnsw_gpu_shim_attach(HnswMetadata *meta, hnsw_gpu_index *ix);
int hnsw_gpu_shim_detach(HnswMetadata *meta);

/* Without an attached mirror the four symbols run on a validated cache (csrc/shim_cache.h): a mirror kept across
 * calls whose every answer is checked against the host's pages along the walk that produced it, and patched where the
 * host has changed.  Counters of the calling thread's cache: snapshots (full walks), searches, search rounds, inserts,
 * insert rounds, elements patched, fallbacks to a full walk, elements read for validation.  PG_EMBEDDING_GPU_CACHE=0
 * switches the cache off (every call re-mirrors the index); PG_EMBEDDING_GPU_CACHE_MAX_MB (default 16384) bounds the host
 * memory one cached index may keep (its flat image, N x element size): a larger index is mirrored per call.
 * The cache belongs to the calling thread (a Postgres backend is one): at most four indexes, least recently used first out;
 * a host that ends threads calls hnsw_gpu_shim_cache_clear() on them first, or their mirrors stay allocated. */
void hnsw_gpu_shim_cache_stats(uint64_t out[8]);
/* Where this thread's hnsw_bind_point calls spent their time, cumulative nanoseconds: out[0] the validated cache's preparation
 * (pick + traced validation walk + host-side comparison), [1] the device insert (hnsw_gpu_index_insert_one), [2] the write-back
 * through hnsw_begin_write, [3] everything else, [4] = number of calls. */
void hnsw_gpu_shim_insert_times(uint64_t out[5]);
void hnsw_gpu_shim_cache_clear(void);

#ifdef __cplusplus
}
#endif
#endif
