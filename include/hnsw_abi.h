/*
 * hnsw_abi.h — the C boundary between the Postgres glue (embedding.c) and the
 * HNSW hot path, restated so that the MI355X library is a link-level drop-in.
 *
 * This header declares, with identical names / argument order / types / struct
 * layout, everything the reference declares in embedding.h:17-56.  It is kept
 * byte-compatible on LP64 (tests/test_abi.py compiles a static_assert TU against
 * the reference header when /root/reference is present).
 *
 * Direction of each symbol:
 *   EXPORTED by libembedding_gpu.so (replaces hnswalg.cpp / distfunc.c):
 *       hnsw_search, hnsw_bind_point, hnsw_dist_func, hnsw_init_dist_func
 *   IMPORTED from the host (embedding.c in Postgres; oracle/flat_host.c in tests):
 *       hnsw_begin_read, hnsw_end_read, hnsw_begin_write, hnsw_end_write,
 *       hnsw_prefetch, hnsw_is_deleted
 */
#ifndef PG_EMBEDDING_AMD_HNSW_ABI_H
#define PG_EMBEDDING_AMD_HNSW_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scalar types — embedding.h:17-20. */
typedef float    coord_t;   /* one vector component                       */
typedef float    dist_t;    /* a distance value                            */
typedef uint32_t idx_t;     /* dense element number inside one index       */
typedef uint64_t label_t;   /* heap TID (6 B) + 16 flag bits, see below    */

/* Metric selector — embedding.h:22-26.  Values are part of the ABI (they index
 * the reference's dispatch table, distfunc.c:157). */
typedef enum {
	DIST_L2        = 0,     /* sqrtf(sum (a-b)^2)        distfunc.c:28-65,121-130 */
	DIST_COSINE    = 1,     /* 1 - a.b / sqrt(|a|^2|b|^2) distfunc.c:133-145       */
	DIST_MANHATTAN = 2      /* sum |a-b|                 distfunc.c:147-155       */
} dist_func_t;

/* Index description handed to every call — embedding.h:28-42.  In Postgres it is
 * the FIRST member of the host's private HnswIndex (embedding.c:65-75), which is
 * how host callbacks recover their own state from the pointer we pass back.
 * Derivation of the size/offset members: embedding.c:222-229. */
typedef struct
{
	size_t		dim;                    /* number of coordinates                               */
	size_t		data_size;              /* dim * sizeof(coord_t)                               */
	size_t		offset_data;            /* (maxM + 1) * sizeof(idx_t): vector starts here      */
	size_t		offset_label;           /* offset_data + data_size: label starts here          */
	size_t		size_data_per_element;  /* offset_label + sizeof(label_t)                      */
	size_t		elems_per_page;         /* host paging detail; unused by the hot path          */
	size_t		M;                      /* links chosen per insert                             */
	size_t		maxM;                   /* link-list capacity = 2*M (embedding.c:224)          */
	size_t		efConstruction;         /* beam width used by hnsw_bind_point                  */
	size_t		efSearch;               /* beam width AND result count of hnsw_search          */
	idx_t		enterpoint_node;        /* always 0 in the reference (embedding.c:235)         */
	dist_func_t dist_func;
} HnswMetadata;

/* Element image shared with the host (embedding.c:222-228, 619-621):
 *     [u32 count][u32 link * maxM][f32 * dim][u64 label]
 * label = { 6-byte ItemPointer, u16 flags }; flags bit 0 (= bit 48 of the
 * little-endian u64) marks a vacuumed row (embedding.c:44,50-56,948-953). */
#define HNSW_LABEL_DELETED_BIT 48

/* ---- exported by the hot-path library ------------------------------------ */

/* k-NN search with k = ef = meta->efSearch (embedding.h:46, hnswalg.cpp:256-277).
 * On success returns true, *n_results <= meta->efSearch and *results = a
 * malloc()ed array (caller free()s, embedding.c:327) of labels ascending by
 * (distance, label), vacuumed labels removed.  Empty index: true, 0 results.
 * Any internal failure (including a HIP error): false, nothing allocated. */
extern bool hnsw_search(HnswMetadata* meta, const coord_t *point, size_t* n_results, label_t** results);

/* Link element `idx` (already stored, zero-linked, by the host) into the graph
 * (embedding.h:47, hnswalg.cpp:279-291).  false on failure. */
extern bool hnsw_bind_point(HnswMetadata* meta, const coord_t *point, idx_t idx);

/* One distance (embedding.h:55, distfunc.c:171-174). */
extern dist_t hnsw_dist_func(dist_func_t dist, coord_t const* ax, coord_t const* bx, size_t dim);

/* Once-per-process initialisation hook (embedding.h:56, distfunc.c:159-169;
 * called from _PG_init, embedding.c:150). */
extern void   hnsw_init_dist_func(void);

/* ---- imported from the host ----------------------------------------------- */

/* Pin element idx and expose pointers into it; any out-pointer may be NULL.
 * Returns false when idx does not exist (embedding.c:704-757).  Pointers stay
 * valid until the matching hnsw_end_read; pins nest LIFO, at most 4 deep
 * (embedding.c:40,714-715). */
extern bool hnsw_begin_read(HnswMetadata* meta, idx_t idx, idx_t** indexes, coord_t** coords, label_t* label);
extern void hnsw_end_read(HnswMetadata* meta);
/* Same for modification; at most one write pin (embedding.c:769-843). */
extern void hnsw_begin_write(HnswMetadata* meta, idx_t idx, idx_t** indexes, coord_t** coords, label_t* label);
extern void hnsw_end_write(HnswMetadata* meta);
/* Advisory (embedding.c:845-850). */
extern void hnsw_prefetch(HnswMetadata* meta, idx_t idx);
/* Flag test on a label (embedding.c:948-953). */
extern bool hnsw_is_deleted(label_t label);

#ifdef __cplusplus
}
#endif
#endif /* PG_EMBEDDING_AMD_HNSW_ABI_H */
