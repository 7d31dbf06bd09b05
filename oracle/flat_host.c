/*
 * oracle/flat_host.c — TEST INFRASTRUCTURE ONLY.
 *
 * A flat-memory stand-in for the Postgres storage adapter of the reference
 * (embedding.c:594-850, 948-953): it supplies the six storage callbacks the HNSW
 * hot path imports through embedding.h:44-53 over ONE contiguous array of
 * elements laid out exactly like an index tuple (embedding.c:222-228):
 *
 *     [u32 count][u32 link * maxM][f32 * dim][u64 label]
 *
 * It is linked two ways (oracle/Makefile):
 *   oracle/_ref/libpgemb_ref.so   with the UNMODIFIED reference distfunc.c and
 *                                 hnswalg.cpp  -> the real reference hot path;
 *   oracle/_build/libflat_host.so alone -> the host side for the product shim
 *                                 (libembedding_gpu.so) in the drop-in tests.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * Like the reference host it relies on HnswMetadata being the first member of
 * its private index struct (embedding.c:65-75,706).
 */
#ifdef ORACLE_REF_BUILD
#include "postgres.h"
#include "embedding.h"      /* the reference's own header, from /root/reference */
#else
#include "hnsw_abi.h"
#endif

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct FlatIndex
{
	HnswMetadata meta;      /* MUST stay first: callbacks down-cast the meta pointer */
	char   *data;
	size_t  n;              /* elements stored                                        */
	size_t  cap;            /* elements allocated                                     */
	int     owns_data;
	size_t  page_real;      /* 0 = dense idx space; else elements that really fit a page:
							 * idx = blk*elems_per_page + off with off < page_real, i.e. the
							 * tail-of-page holes of embedding.c:229,693     */
} FlatIndex;

/* Per-thread instrumentation (E_q = coords reads, H_q = link reads). */
static __thread uint64_t tl_coords_reads;
static __thread uint64_t tl_link_reads;
static __thread uint64_t tl_label_reads;
static __thread int      tl_pin_depth;
static __thread int      tl_max_pin_depth;

#define FLAT_MAX_PINS 4    /* HNSW_STACK_SIZE, embedding.c:40 */

/* element number -> storage slot; false when the element does not exist */
static inline bool idx_to_slot(const FlatIndex *f, idx_t idx, size_t *slot)
{
	if (!f->page_real) { *slot = idx; return (size_t) idx < f->n; }
	size_t epp = f->meta.elems_per_page, blk = idx / epp, off = idx % epp;
	if (off >= f->page_real) return false;
	*slot = blk * f->page_real + off;
	return *slot < f->n;
}
static inline idx_t slot_to_idx(const FlatIndex *f, size_t slot)
{
	if (!f->page_real) return (idx_t) slot;
	return (idx_t) ((slot / f->page_real) * f->meta.elems_per_page + slot % f->page_real);
}
static inline char *slot_ptr(FlatIndex *f, size_t slot)
{
	return f->data + slot * f->meta.size_data_per_element;
}

/* ------------------------------------------------------------------------- */
/* The six imported callbacks.                                               */
/* ------------------------------------------------------------------------- */

/* embedding.c:704-757: false when the element does not exist. */
bool hnsw_begin_read(HnswMetadata *meta, idx_t idx, idx_t **indexes, coord_t **coords, label_t *label)
{
	FlatIndex *f = (FlatIndex *) meta;
	char *p;
	size_t slot;

	if (!idx_to_slot(f, idx, &slot))
		return false;
	if (tl_pin_depth >= FLAT_MAX_PINS)
	{
		/* The Postgres host raises elog(ERROR) here (embedding.c:714-715). */
		fprintf(stderr, "flat_host: more than %d nested read pins\n", FLAT_MAX_PINS);
		abort();
	}
	tl_pin_depth++;
	if (tl_pin_depth > tl_max_pin_depth)
		tl_max_pin_depth = tl_pin_depth;
	p = slot_ptr(f, slot);
	if (indexes) { *indexes = (idx_t *) p; tl_link_reads++; }
	if (coords)  { *coords = (coord_t *) (p + meta->offset_data); tl_coords_reads++; }
	if (label)   { memcpy(label, p + meta->offset_label, sizeof(label_t)); tl_label_reads++; }
	return true;
}

void hnsw_end_read(HnswMetadata *meta)
{
	(void) meta;
	if (tl_pin_depth <= 0)
	{
		fprintf(stderr, "flat_host: hnsw_end_read without a pin\n");
		abort();
	}
	tl_pin_depth--;
}

static __thread int tl_write_pins;

/* embedding.c:769-821. */
void hnsw_begin_write(HnswMetadata *meta, idx_t idx, idx_t **indexes, coord_t **coords, label_t *label)
{
	FlatIndex *f = (FlatIndex *) meta;
	char *p;
	size_t slot;

	if (!idx_to_slot(f, idx, &slot) || tl_write_pins != 0)
	{
		fprintf(stderr, "flat_host: bad hnsw_begin_write(%u)\n", (unsigned) idx);
		abort();
	}
	tl_write_pins = 1;
	p = slot_ptr(f, slot);
	if (indexes) *indexes = (idx_t *) p;
	if (coords)  *coords = (coord_t *) (p + meta->offset_data);
	if (label)   memcpy(label, p + meta->offset_label, sizeof(label_t));
}

void hnsw_end_write(HnswMetadata *meta)
{
	(void) meta;
	tl_write_pins = 0;
}

/* embedding.c:845-850. */
void hnsw_prefetch(HnswMetadata *meta, idx_t idx)
{
	FlatIndex *f = (FlatIndex *) meta;
	size_t slot;
	if (idx_to_slot(f, idx, &slot))
		__builtin_prefetch(slot_ptr(f, slot) + meta->offset_data);
}

/* embedding.c:948-953: flags half-word bit 0 == bit 48 of the u64. */
bool hnsw_is_deleted(label_t label)
{
	return ((label >> 48) & 1u) != 0;
}

/* ------------------------------------------------------------------------- */
/* Host-side API used by the Python test drivers (ctypes).                   */
/* ------------------------------------------------------------------------- */

/* Mirrors hnsw_get_index, embedding.c:214-244 (BLCKSZ 8192, 24-byte page header,
 * 4-byte opaque, 4-byte line pointer). */
static void fill_meta(HnswMetadata *m, size_t dim, size_t M, size_t efc, size_t efs, int dist_func)
{
	m->dim = dim;
	m->M = M;
	m->maxM = M * 2;
	m->data_size = dim * sizeof(coord_t);
	m->offset_data = (m->maxM + 1) * sizeof(idx_t);
	m->offset_label = m->offset_data + m->data_size;
	m->size_data_per_element = m->offset_label + sizeof(label_t);
	m->elems_per_page = (8192 - 24 - 4) / (m->size_data_per_element + 4);
	m->efConstruction = efc;
	m->efSearch = efs;
	m->dist_func = (dist_func_t) dist_func;
	m->enterpoint_node = 0;
}

FlatIndex *flat_create(size_t dim, size_t M, size_t efc, size_t efs, int dist_func, size_t capacity)
{
	FlatIndex *f = (FlatIndex *) calloc(1, sizeof(FlatIndex));
	if (!f)
		return NULL;
	fill_meta(&f->meta, dim, M, efc, efs, dist_func);
	f->cap = capacity ? capacity : 16;
	f->data = (char *) calloc(f->cap, f->meta.size_data_per_element);
	f->owns_data = 1;
	if (!f->data) { free(f); return NULL; }
	hnsw_init_dist_func();
	return f;
}

void flat_destroy(FlatIndex *f)
{
	if (!f) return;
	if (f->owns_data) free(f->data);
	free(f);
}

/* Emulate a page that really holds `page_real` < elems_per_page elements (MAXALIGN padding,
 * embedding.c:229 vs PageAddItem): element numbers get holes at every page tail.  Must be
 * called on an empty index. */
int flat_set_page_real(FlatIndex *f, size_t page_real)
{
	if (f->n != 0 || page_real == 0 || page_real > f->meta.elems_per_page) return -1;
	f->page_real = page_real == f->meta.elems_per_page ? 0 : page_real;
	return 0;
}
/* highest element number + 1 */
size_t flat_idx_end(FlatIndex *f) { return f->n ? (size_t) slot_to_idx(f, f->n - 1) + 1 : 0; }

HnswMetadata *flat_meta(FlatIndex *f)   { return &f->meta; }
size_t        flat_count(FlatIndex *f)  { return f->n; }
void         *flat_data(FlatIndex *f)   { return f->data; }
size_t        flat_elem_size(FlatIndex *f) { return f->meta.size_data_per_element; }
void          flat_set_ef_search(FlatIndex *f, size_t efs) { f->meta.efSearch = efs; }
void          flat_set_ef_construction(FlatIndex *f, size_t efc) { f->meta.efConstruction = efc; }

static int flat_reserve(FlatIndex *f, size_t want)
{
	if (want <= f->cap) return 0;
	size_t ncap = f->cap * 2;
	if (ncap < want) ncap = want;
	char *nd = (char *) realloc(f->data, ncap * f->meta.size_data_per_element);
	if (!nd) return -1;
	memset(nd + f->cap * f->meta.size_data_per_element, 0,
		   (ncap - f->cap) * f->meta.size_data_per_element);
	f->data = nd;
	f->cap = ncap;
	return 0;
}

/* Store one zero-linked element, as hnsw_add_point does before binding
 * (embedding.c:619-621,670).  Returns its idx, or -1. */
long flat_append(FlatIndex *f, const coord_t *vec, label_t label)
{
	if (flat_reserve(f, f->n + 1) != 0) return -1;
	char *p = slot_ptr(f, f->n);
	memset(p, 0, f->meta.offset_data);
	memcpy(p + f->meta.offset_data, vec, f->meta.data_size);
	memcpy(p + f->meta.offset_label, &label, sizeof(label));
	return (long) slot_to_idx(f, f->n++);
}

/* Insert = append + hnsw_bind_point (embedding.c:606-701 minus paging/WAL). */
long flat_add(FlatIndex *f, const coord_t *vec, label_t label)
{
	long idx = flat_append(f, vec, label);
	if (idx < 0) return -1;
	if (!hnsw_bind_point(&f->meta, vec, (idx_t) idx))
		return -2;
	return idx;
}

long flat_add_many(FlatIndex *f, const coord_t *vecs, const label_t *labels, size_t n)
{
	for (size_t i = 0; i < n; i++)
	{
		long r = flat_add(f, vecs + i * f->meta.dim, labels ? labels[i] : (label_t) slot_to_idx(f, f->n));
		if (r < 0) return r;
	}
	return (long) f->n;
}

/* Install a ready-made element image (e.g. a graph built elsewhere) so that the
 * CPU path searches the identical bytes. */
int flat_load_raw(FlatIndex *f, const void *bytes, size_t n)
{
	if (flat_reserve(f, n) != 0) return -1;
	memcpy(f->data, bytes, n * f->meta.size_data_per_element);
	f->n = n;
	return 0;
}

/* bulkdelete analogue: set / clear DELETED_FLAG in the label (embedding.c:920-926). */
void flat_set_deleted(FlatIndex *f, idx_t idx, int deleted)
{
	label_t l;
	size_t slot;
	if (!idx_to_slot(f, idx, &slot)) return;
	char *p = slot_ptr(f, slot) + f->meta.offset_label;
	memcpy(&l, p, sizeof(l));
	if (deleted) l |= ((label_t) 1 << 48); else l &= ~((label_t) 1 << 48);
	memcpy(p, &l, sizeof(l));
}

dist_t flat_dist(int func, const coord_t *a, const coord_t *b, size_t dim)
{
	hnsw_init_dist_func();
	return hnsw_dist_func((dist_func_t) func, a, b, dim);
}

void flat_dist_many(int func, const coord_t *q, const coord_t *rows, size_t nrows, size_t dim, dist_t *out)
{
	hnsw_init_dist_func();
	for (size_t i = 0; i < nrows; i++)
		out[i] = hnsw_dist_func((dist_func_t) func, q, rows + i * dim, dim);
}

void flat_counters_reset(void)
{
	tl_coords_reads = tl_link_reads = tl_label_reads = 0;
	tl_max_pin_depth = 0;
}
void flat_counters_get(uint64_t out[4])
{
	out[0] = tl_coords_reads; out[1] = tl_link_reads; out[2] = tl_label_reads;
	out[3] = (uint64_t) tl_max_pin_depth;
}

/* One search through the boundary symbol.  out must hold efs labels.
 * Returns 0 / -1 (hnsw_search returned false). */
int flat_search(FlatIndex *f, const coord_t *q, size_t efs, label_t *out, size_t *n_out)
{
	size_t n = 0;
	label_t *res = NULL;
	f->meta.efSearch = efs;
	if (!hnsw_search(&f->meta, q, &n, &res))
		return -1;
	if (n) memcpy(out, res, n * sizeof(label_t));
	free(res);
	*n_out = n;
	return 0;
}

typedef struct
{
	FlatIndex     *f;
	const coord_t *Q;
	size_t         q0, q1, efs;
	label_t       *labels;
	uint32_t      *counts;
	uint32_t      *evals;
	uint32_t      *hops;
	int            failed;
} SearchJob;

static void *search_worker(void *arg)
{
	SearchJob *j = (SearchJob *) arg;
	size_t dim = j->f->meta.dim;
	for (size_t q = j->q0; q < j->q1; q++)
	{
		size_t n = 0;
		label_t *res = NULL;
		uint64_t c0 = tl_coords_reads, l0 = tl_link_reads;
		if (!hnsw_search(&j->f->meta, j->Q + q * dim, &n, &res)) { j->failed = 1; continue; }
		if (j->labels && n) memcpy(j->labels + q * j->efs, res, n * sizeof(label_t));
		free(res);
		if (j->counts) j->counts[q] = (uint32_t) n;
		if (j->evals)  j->evals[q] = (uint32_t) (tl_coords_reads - c0);
		if (j->hops)   j->hops[q] = (uint32_t) (tl_link_reads - l0);
	}
	return NULL;
}

/* nq searches over `nthreads` host threads (one query per thread at a time, index
 * shared read-only).  Returns wall seconds of the query loop, or a negative value. */
double flat_search_many(FlatIndex *f, const coord_t *Q, size_t nq, size_t efs, int nthreads,
						label_t *labels, uint32_t *counts, uint32_t *evals, uint32_t *hops)
{
	if (nthreads < 1) nthreads = 1;
	if ((size_t) nthreads > nq && nq > 0) nthreads = (int) nq;
	f->meta.efSearch = efs;
	SearchJob *jobs = (SearchJob *) calloc((size_t) nthreads, sizeof(SearchJob));
	pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
	struct timespec t0, t1;
	int failed = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int t = 0; t < nthreads; t++)
	{
		jobs[t].f = f; jobs[t].Q = Q; jobs[t].efs = efs;
		jobs[t].q0 = nq * (size_t) t / (size_t) nthreads;
		jobs[t].q1 = nq * (size_t) (t + 1) / (size_t) nthreads;
		jobs[t].labels = labels; jobs[t].counts = counts; jobs[t].evals = evals; jobs[t].hops = hops;
		if (nthreads == 1) search_worker(&jobs[t]);
		else pthread_create(&th[t], NULL, search_worker, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++)
	{
		if (nthreads > 1) pthread_join(th[t], NULL);
		failed |= jobs[t].failed;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(jobs); free(th);
	if (failed) return -1.0;
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
