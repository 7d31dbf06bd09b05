/*
 * tests/double/engine_double.c — TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for the part of libhnsw_gpu.so's C API (include/hnsw_gpu.h) that hnsw_gpu_server
 * calls, implemented on the CPU oracle (oracle/hnsw_port.c).  tests/server_util.py links the
 * server's own source (pg_embedding_amd/csrc/server_main.cpp) against this file INSTEAD of the
 * HIP library, into tests/_build/hnsw_gpu_server_double, so that the server's protocol, batching,
 * locking and the client library can be tested in the CPU-only container.  The product binary
 * (pg_embedding_amd/bin/hnsw_gpu_server) links libhnsw_gpu.so and has no switch to get here.
 *
 * HGS_DOUBLE_SLEEP_US makes every search batch take at least that long, so that requests pile up
 * behind it the way they do behind a busy device.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hnsw_gpu.h"

/* oracle/hnsw_port.c */
typedef struct PortIndex PortIndex;
PortIndex *port_create(size_t dim, size_t M, size_t efc, size_t efs, int func, size_t capacity);
void   port_destroy(PortIndex *ix);
size_t port_count(PortIndex *ix);
void  *port_data(PortIndex *ix);
size_t port_elem_size(PortIndex *ix);
int    port_load_raw(PortIndex *ix, const void *bytes, size_t n);
void   port_set_deleted(PortIndex *ix, uint32_t idx, int deleted);
int    port_search(PortIndex *ix, const float *q, size_t ef, uint64_t *label_out, float *dist_out,
				   size_t *n_out, uint32_t *evals, uint32_t *hops);
int    port_bind_point(PortIndex *ix, const float *point, uint32_t cur_c);
float  port_dist(int func, const float *q, const float *x, size_t dim);

struct hnsw_gpu_index { PortIndex *p; HnswMetadata meta; size_t charged; };
struct hnsw_gpu_ctx
{
	hnsw_gpu_index *ix;
	/* the "launch" of hnsw_gpu_search_batch_ctx_flags: a thread that walks the queries one by one */
	volatile int busy;
	const coord_t *q; size_t nq, ef; label_t *labels; dist_t *dists; uint32_t *counts; uint32_t *done;
};

static __thread char t_err[256] = "";
const char *hnsw_gpu_last_error(void) { return t_err; }
int hnsw_gpu_device_count(void) { return 1; }

/* HGS_DOUBLE_CAPACITY = elements the "device" holds over all mirrors: beyond it an upload fails with
 * HNSW_GPU_ERR_NOMEM like a failed hipMalloc, which is what makes the server evict idle mirrors. */
static size_t g_live_elements = 0;

int hnsw_gpu_index_create_from_flat(const HnswMetadata *meta, const void *elements, size_t n, int device,
									hnsw_gpu_index **out)
{
	(void) device;
	const char *cap = getenv("HGS_DOUBLE_CAPACITY");
	if (cap && __atomic_load_n(&g_live_elements, __ATOMIC_RELAXED) + n > (size_t) atol(cap))
	{
		snprintf(t_err, sizeof(t_err), "double: out of device memory");
		return HNSW_GPU_ERR_NOMEM;
	}
	hnsw_gpu_index *ix = (hnsw_gpu_index *) calloc(1, sizeof(*ix));
	if (!ix) return HNSW_GPU_ERR_NOMEM;
	ix->meta = *meta;
	ix->p = port_create(meta->dim, meta->M, meta->efConstruction, meta->efSearch, (int) meta->dist_func, n ? n : 16);
	if (!ix->p || (n && port_load_raw(ix->p, elements, n) != 0))
	{
		snprintf(t_err, sizeof(t_err), "double: cannot load %zu elements", n);
		if (ix->p) port_destroy(ix->p);
		free(ix);
		return HNSW_GPU_ERR_NOMEM;
	}
	ix->charged = n;
	__atomic_fetch_add(&g_live_elements, n, __ATOMIC_RELAXED);
	*out = ix;
	return HNSW_GPU_OK;
}

void hnsw_gpu_index_destroy(hnsw_gpu_index *ix)
{
	if (!ix) return;
	__atomic_fetch_sub(&g_live_elements, ix->charged, __ATOMIC_RELAXED);
	port_destroy(ix->p);
	free(ix);
}

size_t hnsw_gpu_index_count(const hnsw_gpu_index *ix) { return ix ? port_count(ix->p) : 0; }

int hnsw_gpu_index_update_from_flat(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count)
{
	const size_t esz = port_elem_size(ix->p), have = port_count(ix->p);
	if (first > have) { snprintf(t_err, sizeof(t_err), "double: first %zu > count %zu", first, have); return HNSW_GPU_ERR_ARG; }
	const size_t total = first + count > have ? first + count : have;
	char *img = (char *) malloc(total * esz ? total * esz : 1);
	if (!img) return HNSW_GPU_ERR_NOMEM;
	memcpy(img, port_data(ix->p), have * esz);
	memcpy(img + first * esz, elements, count * esz);
	int rc = port_load_raw(ix->p, img, total);
	free(img);
	return rc == 0 ? HNSW_GPU_OK : HNSW_GPU_ERR_NOMEM;
}

int hnsw_gpu_index_reserve(hnsw_gpu_index *ix, size_t capacity) { (void) ix; (void) capacity; return HNSW_GPU_OK; }
size_t hnsw_gpu_index_capacity(const hnsw_gpu_index *ix) { (void) ix; return (size_t) 1 << 40; }

int hnsw_gpu_index_append(hnsw_gpu_index *ix, const coord_t *vectors, const label_t *labels, size_t n)
{
	const size_t esz = port_elem_size(ix->p), have = port_count(ix->p), dim = ix->meta.dim;
	char *img = (char *) calloc(n ? n : 1, esz);
	if (!img) return HNSW_GPU_ERR_NOMEM;
	for (size_t i = 0; i < n; i++)
	{
		label_t l = labels ? labels[i] : (label_t) (have + i);
		memcpy(img + i * esz + ix->meta.offset_data, vectors + i * dim, dim * 4);
		memcpy(img + i * esz + ix->meta.offset_label, &l, 8);
	}
	int rc = hnsw_gpu_index_update_from_flat(ix, img, have, n);
	free(img);
	return rc;
}

int hnsw_gpu_index_link(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio, void *stream)
{
	(void) max_batch; (void) ratio; (void) stream;       /* always the reference's serial order */
	const size_t esz = port_elem_size(ix->p);
	for (size_t i = first; i < first + count; i++)
	{
		const float *v = (const float *) ((char *) port_data(ix->p) + i * esz + ix->meta.offset_data);
		if (port_bind_point(ix->p, v, (uint32_t) i) != 0) return HNSW_GPU_ERR_INTERNAL;
	}
	return HNSW_GPU_OK;
}

int hnsw_gpu_index_get_links(hnsw_gpu_index *ix, idx_t idx, idx_t *out)
{
	if (idx >= port_count(ix->p)) return HNSW_GPU_ERR_ARG;
	memcpy(out, (char *) port_data(ix->p) + (size_t) idx * port_elem_size(ix->p), (ix->meta.maxM + 1) * 4);
	return HNSW_GPU_OK;
}

int hnsw_gpu_index_get_link_lists(hnsw_gpu_index *ix, idx_t idx, idx_t *mine, idx_t *others)
{
	const size_t maxM = ix->meta.maxM;
	int rc = hnsw_gpu_index_get_links(ix, idx, mine);
	for (uint32_t j = 0; rc == HNSW_GPU_OK && j < mine[0]; j++) rc = hnsw_gpu_index_get_links(ix, mine[1 + j], others + (size_t) j * (maxM + 1));
	return rc;
}

int hnsw_gpu_index_insert_one(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, idx_t *mine, idx_t *others)
{
	if ((size_t) idx != port_count(ix->p)) return HNSW_GPU_ERR_ARG;
	int rc = hnsw_gpu_index_append(ix, point, &label, 1);
	if (rc == HNSW_GPU_OK && idx > 0) rc = hnsw_gpu_index_link(ix, idx, 1, 1, 0, NULL);
	if (rc == HNSW_GPU_OK) rc = hnsw_gpu_index_get_link_lists(ix, idx, mine, others);
	return rc;
}

int hnsw_gpu_index_insert_candidates(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx,
									 const dist_t *cand_dist, uint32_t ncand, idx_t *mine, idx_t *others)
{
	(void) cand_idx; (void) cand_dist; (void) ncand;     /* the double walks itself: same graph either way */
	return hnsw_gpu_index_insert_one(ix, point, label, idx, mine, others);
}

int hnsw_gpu_index_export_flat(hnsw_gpu_index *ix, void *elements)
{
	memcpy(elements, port_data(ix->p), port_count(ix->p) * port_elem_size(ix->p));
	return HNSW_GPU_OK;
}

int hnsw_gpu_index_set_deleted(hnsw_gpu_index *ix, idx_t idx, int deleted)
{
	if (idx >= port_count(ix->p)) return HNSW_GPU_ERR_ARG;
	port_set_deleted(ix->p, idx, deleted);
	return HNSW_GPU_OK;
}

int hnsw_gpu_index_set_deleted_batch(hnsw_gpu_index *ix, const idx_t *idx, size_t count, int deleted)
{
	for (size_t i = 0; i < count; i++)
		if (idx[i] >= port_count(ix->p)) return HNSW_GPU_ERR_ARG;
	for (size_t i = 0; i < count; i++) port_set_deleted(ix->p, idx[i], deleted);
	return HNSW_GPU_OK;
}

/* host-pointer searches: what libembedding_gpu.so (embedding_shim.cpp) calls; tests/server_util.py links that source
 * against this file for the CPU tests of its validated mirror cache */
int port_search_trace(PortIndex *ix, const float *q, size_t ef, int base, uint64_t *label_out, float *dist_out,
					  size_t *n_out, uint32_t *evals, uint32_t *pops, size_t pops_cap, uint32_t *npops);

int hnsw_gpu_search_batch(hnsw_gpu_index *ix, const coord_t *queries, size_t nq, size_t ef, label_t *labels, dist_t *dists,
						  uint32_t *counts)
{
	float *d = (float *) malloc((ef ? ef : 1) * sizeof(float));
	if (!d) return HNSW_GPU_ERR_NOMEM;
	for (size_t i = 0; i < nq; i++)
	{
		size_t n = 0;
		for (size_t k = 0; k < ef; k++) labels[i * ef + k] = ~(label_t) 0;
		if (port_search(ix->p, queries + i * ix->meta.dim, ef, labels + i * ef, d, &n, NULL, NULL) != 0) { free(d); return HNSW_GPU_ERR_INTERNAL; }
		if (dists) memcpy(dists + i * ef, d, n * sizeof(float));
		counts[i] = (uint32_t) n;
	}
	free(d);
	return HNSW_GPU_OK;
}

int hnsw_gpu_search_trace(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, label_t *labels, dist_t *dists,
						  uint32_t *count, idx_t *pops, size_t pops_cap, uint32_t *npops, uint32_t *nevals)
{
	/* as on the device, a beam wider than the index is a beam of the index size */
	const size_t have = port_count(ix->p), eff = ef < (have ? have : 1) ? ef : (have ? have : 1);
	float *d = (float *) malloc((ef ? ef : 1) * sizeof(float));
	uint64_t *l = (uint64_t *) malloc((ef ? ef : 1) * sizeof(uint64_t));
	if (!d || !l) { free(d); free(l); return HNSW_GPU_ERR_NOMEM; }
	size_t n = 0;
	int rc = port_search_trace(ix->p, query, eff, base, l, d, &n, nevals, pops, pops_cap, npops);
	if (rc == 0)
	{
		memcpy(labels, l, n * sizeof(uint64_t));
		if (dists) memcpy(dists, d, n * sizeof(float));
		*count = (uint32_t) n;
	}
	free(d); free(l);
	return rc == 0 ? HNSW_GPU_OK : HNSW_GPU_ERR_INTERNAL;
}

/* the three-step form: the double walks at _begin and hands the sequence out in slices, so that the caller's incremental
 * consumption is exercised */
static __thread struct { hnsw_gpu_index *ix; size_t ef, cap, seen; int base; uint64_t *lab; float *dst; uint32_t *pops; uint32_t cnt, npops, nev; } t_tr;

int hnsw_gpu_search_trace_begin(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, size_t pops_cap)
{
	free(t_tr.lab); free(t_tr.dst); free(t_tr.pops);
	memset(&t_tr, 0, sizeof(t_tr));
	t_tr.lab = (uint64_t *) malloc((ef ? ef : 1) * 8); t_tr.dst = (float *) malloc((ef ? ef : 1) * 4); t_tr.pops = (uint32_t *) malloc(pops_cap * 4);
	if (!t_tr.lab || !t_tr.dst || !t_tr.pops) return HNSW_GPU_ERR_NOMEM;
	t_tr.ix = ix; t_tr.ef = ef; t_tr.cap = pops_cap; t_tr.base = base;
	return hnsw_gpu_search_trace(ix, query, ef, base, t_tr.lab, t_tr.dst, &t_tr.cnt, t_tr.pops, pops_cap, &t_tr.npops, &t_tr.nev);
}

int hnsw_gpu_search_trace_poll(hnsw_gpu_index *ix, idx_t *pops, size_t max, size_t *got, int *finished)
{
	if (ix != t_tr.ix) return HNSW_GPU_ERR_ARG;
	const size_t have = t_tr.npops < t_tr.cap ? t_tr.npops : t_tr.cap;
	size_t k = have - t_tr.seen;
	if (k > max) k = max;
	if (k > 7) k = 7;                                   /* slices: several polls per walk */
	memcpy(pops, t_tr.pops + t_tr.seen, k * 4);
	t_tr.seen += k;
	*got = k;
	*finished = t_tr.seen == have;
	return HNSW_GPU_OK;
}

int hnsw_gpu_search_trace_end(hnsw_gpu_index *ix, label_t *labels, dist_t *dists, uint32_t *count, uint32_t *npops, uint32_t *nevals)
{
	if (ix != t_tr.ix) return HNSW_GPU_ERR_ARG;
	memcpy(labels, t_tr.lab, t_tr.cnt * 8);
	if (dists) memcpy(dists, t_tr.dst, t_tr.cnt * 4);
	*count = t_tr.cnt; *npops = t_tr.npops;
	if (nevals) *nevals = t_tr.nev;
	t_tr.ix = NULL;
	return HNSW_GPU_OK;
}

int hnsw_gpu_ctx_create(hnsw_gpu_index *ix, hnsw_gpu_ctx **out)
{
	hnsw_gpu_ctx *c = (hnsw_gpu_ctx *) calloc(1, sizeof(*c));
	if (!c) return HNSW_GPU_ERR_NOMEM;
	c->ix = ix;
	*out = c;
	return HNSW_GPU_OK;
}

void hnsw_gpu_ctx_destroy(hnsw_gpu_ctx *c)
{
	if (!c) return;
	while (c->busy) { struct timespec ts = { 0, 100000 }; nanosleep(&ts, NULL); }
	free(c);
}

int hnsw_gpu_search_batch_ctx_host(hnsw_gpu_ctx *c, const coord_t *queries, size_t nq, size_t ef, label_t *labels,
								   dist_t *dists, uint32_t *counts)
{
	const char *fail_at = getenv("HGS_DOUBLE_FAIL_EF");       /* error-path tests */
	if (fail_at && (size_t) atol(fail_at) == ef)
	{
		snprintf(t_err, sizeof(t_err), "double: asked to fail at ef %zu", ef);
		return HNSW_GPU_ERR_INTERNAL;
	}
	const char *us = getenv("HGS_DOUBLE_SLEEP_US");
	if (us && atol(us) > 0)
	{
		struct timespec ts = { atol(us) / 1000000, (atol(us) % 1000000) * 1000 };
		nanosleep(&ts, NULL);
	}
	const size_t dim = c->ix->meta.dim;
	for (size_t q = 0; q < nq; q++)
	{
		size_t n = 0;
		for (size_t i = 0; i < ef; i++) { labels[q * ef + i] = ~(label_t) 0; if (dists) dists[q * ef + i] = 1.0f / 0.0f; }
		port_search(c->ix->p, queries + q * dim, ef, labels + q * ef, dists ? dists + q * ef : NULL, &n, NULL, NULL);
		counts[q] = (uint32_t) n;
	}
	return HNSW_GPU_OK;
}

int hnsw_gpu_ctx_search_ms(hnsw_gpu_ctx *c, unsigned back, float *ms) { (void) c; (void) back; *ms = 0.f; return HNSW_GPU_OK; }
static unsigned g_walkers_last = 0, g_walkers_max = 0;     /* what the server asked for (tests read them through the stats of the double) */
int hnsw_gpu_ctx_set_walkers(hnsw_gpu_ctx *c, unsigned per_block)
{
	(void) c;
	__atomic_store_n(&g_walkers_last, per_block, __ATOMIC_RELAXED);          /* (several dispatcher lanes call this at once) */
	unsigned m = __atomic_load_n(&g_walkers_max, __ATOMIC_RELAXED);
	while (per_block > m && !__atomic_compare_exchange_n(&g_walkers_max, &m, per_block, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
	return HNSW_GPU_OK;
}
unsigned engine_double_walkers_max(void) { return __atomic_load_n(&g_walkers_max, __ATOMIC_RELAXED); }
/* Streams (include/hnsw_gpu.h): the resident launch is played by threads — 3 x walkers of them, the "walking waves" of the three
 * walking blocks of this 4-block "device" — that take tickets, wait until the host has published past their ticket, answer the ring
 * slot through the oracle and raise its flag; walks end out of order (HGS_DOUBLE_SLEEP_US stretches every third one).  What the server
 * does with a stream — lock-free producers, ring reuse, answer threads, opening / closing / re-shaping sessions, mirror changes and
 * shutdown while one is open — runs in the CPU tier against this.  HGS_DOUBLE_NO_STREAMS=1: opening fails (the server's fallback). */
struct hnsw_gpu_stream
{
	hnsw_gpu_ctx *c;
	size_t ef, ring, dim;
	coord_t *Q; label_t *L; dist_t *D; uint32_t *C; uint32_t *F;
	uint32_t published, next;
	int stop, nthreads, alive;
	pthread_t th[24];
};
static unsigned g_streams_opened = 0;
unsigned engine_double_streams_opened(void) { return __atomic_load_n(&g_streams_opened, __ATOMIC_RELAXED); }

static void *stream_walker(void *arg)
{
	hnsw_gpu_stream *s = (hnsw_gpu_stream *) arg;
	const char *us = getenv("HGS_DOUBLE_SLEEP_US");
	const long stretch = us ? atol(us) : 0;
	for (;;)
	{
		const uint32_t t = __atomic_fetch_add(&s->next, 1u, __ATOMIC_ACQ_REL);
		for (;;)
		{
			if ((int32_t) (__atomic_load_n(&s->published, __ATOMIC_ACQUIRE) - t) > 0) break;
			if (__atomic_load_n(&s->stop, __ATOMIC_ACQUIRE)) { __atomic_fetch_sub(&s->alive, 1, __ATOMIC_ACQ_REL); return NULL; }
			struct timespec ts = { 0, 20000 };
			nanosleep(&ts, NULL);
		}
		const size_t slot = t & (s->ring - 1), ef = s->ef;
		if (stretch > 0 && t % 3 == 0)
		{
			struct timespec ts = { 0, (stretch % 1000000) * 1000 / 4 };
			nanosleep(&ts, NULL);
		}
		size_t n = 0;
		for (size_t i = 0; i < ef; i++) { s->L[slot * ef + i] = ~(label_t) 0; s->D[slot * ef + i] = 1.0f / 0.0f; }
		port_search(s->c->ix->p, s->Q + slot * s->dim, ef, s->L + slot * ef, s->D + slot * ef, &n, NULL, NULL);
		s->C[slot] = (uint32_t) n;
		__atomic_store_n(&s->F[slot], 1u, __ATOMIC_RELEASE);
	}
}

int hnsw_gpu_stream_open(hnsw_gpu_ctx *c, size_t ef, size_t ring, unsigned walkers, hnsw_gpu_stream **out)
{
	if (!c || !out || ef == 0 || ring < 64 || (ring & (ring - 1))) { snprintf(t_err, sizeof(t_err), "double: bad stream arguments"); return HNSW_GPU_ERR_ARG; }
	if (getenv("HGS_DOUBLE_NO_STREAMS")) { snprintf(t_err, sizeof(t_err), "double: streams switched off"); return HNSW_GPU_ERR_INTERNAL; }
	if (ef > 512) { snprintf(t_err, sizeof(t_err), "a stream needs ef <= 512 (the team form of the beam kernel)"); return HNSW_GPU_ERR_ARG; }
	if (c->busy) { snprintf(t_err, sizeof(t_err), "double: context busy"); return HNSW_GPU_ERR_INTERNAL; }
	hnsw_gpu_stream *s = (hnsw_gpu_stream *) calloc(1, sizeof(*s));
	if (!s) return HNSW_GPU_ERR_NOMEM;
	s->c = c; s->ef = ef; s->ring = ring; s->dim = c->ix->meta.dim;
	s->Q = (coord_t *) calloc(ring * s->dim, sizeof(coord_t));
	s->L = (label_t *) calloc(ring * ef, sizeof(label_t));
	s->D = (dist_t *) calloc(ring * ef, sizeof(dist_t));
	s->C = (uint32_t *) calloc(ring, 4);
	s->F = (uint32_t *) calloc(ring, 4);
	if (!s->Q || !s->L || !s->D || !s->C || !s->F) { free(s->Q); free(s->L); free(s->D); free(s->C); free(s->F); free(s); return HNSW_GPU_ERR_NOMEM; }
	if (walkers == 0) walkers = 4;
	if (walkers > 8) walkers = 8;
	s->nthreads = (int) (3 * walkers);
	s->alive = s->nthreads;
	c->busy = 1;                                           /* the context serves the stream until it is closed */
	for (int i = 0; i < s->nthreads; i++) pthread_create(&s->th[i], NULL, stream_walker, s);
	__atomic_fetch_add(&g_streams_opened, 1u, __ATOMIC_RELAXED);
	*out = s;
	return HNSW_GPU_OK;
}

int hnsw_gpu_stream_buffers(hnsw_gpu_stream *s, coord_t **q, label_t **l, dist_t **d, uint32_t **c, uint32_t **f)
{
	if (!s) return HNSW_GPU_ERR_ARG;
	if (q) *q = s->Q;
	if (l) *l = s->L;
	if (d) *d = s->D;
	if (c) *c = s->C;
	if (f) *f = s->F;
	return HNSW_GPU_OK;
}

int hnsw_gpu_stream_publish(hnsw_gpu_stream *s, uint32_t n)
{
	if (!s) return HNSW_GPU_ERR_ARG;
	uint32_t cur = __atomic_load_n(&s->published, __ATOMIC_ACQUIRE);                 /* keeps the maximum, as the library does */
	while ((int32_t) (n - cur) > 0 && !__atomic_compare_exchange_n(&s->published, &cur, n, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) { }
	return HNSW_GPU_OK;
}

int hnsw_gpu_stream_alive(hnsw_gpu_stream *s) { return s && __atomic_load_n(&s->alive, __ATOMIC_ACQUIRE) > 0 && !s->stop; }

static int stream_end(hnsw_gpu_stream *s, int keep_buffers)
{
	if (!s) return HNSW_GPU_OK;
	__atomic_store_n(&s->stop, 1, __ATOMIC_RELEASE);
	for (int i = 0; i < s->nthreads; i++) pthread_join(s->th[i], NULL);
	__atomic_store_n(&s->c->busy, 0, __ATOMIC_RELEASE);
	if (!keep_buffers) { free(s->Q); free(s->L); free(s->D); free(s->C); free(s->F); free(s); }      /* (abandoned: ring AND handle stay, as in the library —
	                                                                                                     a late producer may still publish through it) */
	return HNSW_GPU_OK;
}
int hnsw_gpu_stream_close(hnsw_gpu_stream *s) { return stream_end(s, 0); }
int hnsw_gpu_stream_abandon(hnsw_gpu_stream *s) { return stream_end(s, 1); }      /* (ring and handle stay allocated, as in the library) */
int hnsw_gpu_device_blocks(int device) { (void) device; return 4; }   /* a tiny "device": the server's load policy is exercised with a handful of backends */

static void *flags_worker(void *arg)
{
	hnsw_gpu_ctx *c = (hnsw_gpu_ctx *) arg;
	const size_t dim = c->ix->meta.dim, ef = c->ef;
	const char *us = getenv("HGS_DOUBLE_SLEEP_US");
	for (size_t q = 0; q < c->nq; q++)
	{
		if (us && atol(us) > 0 && q % 8 == 0)                /* walks end at different times */
		{
			struct timespec ts = { 0, (atol(us) % 1000000) * 1000 / 4 };
			nanosleep(&ts, NULL);
		}
		size_t n = 0;
		for (size_t i = 0; i < ef; i++) { c->labels[q * ef + i] = ~(label_t) 0; if (c->dists) c->dists[q * ef + i] = 1.0f / 0.0f; }
		port_search(c->ix->p, c->q + q * dim, ef, c->labels + q * ef, c->dists ? c->dists + q * ef : NULL, &n, NULL, NULL);
		c->counts[q] = (uint32_t) n;
		__atomic_store_n(&c->done[q], 1u, __ATOMIC_RELEASE);
	}
	__atomic_store_n(&c->busy, 0, __ATOMIC_RELEASE);
	return NULL;
}

int hnsw_gpu_search_batch_ctx_flags(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef, label_t *d_labels,
									dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats, uint32_t *d_done)
{
	(void) d_stats;
	const char *fail_at = getenv("HGS_DOUBLE_FAIL_EF");
	if (fail_at && (size_t) atol(fail_at) == ef)
	{
		snprintf(t_err, sizeof(t_err), "double: asked to fail at ef %zu", ef);
		return HNSW_GPU_ERR_INTERNAL;
	}
	if (c->busy) { snprintf(t_err, sizeof(t_err), "double: context busy"); return HNSW_GPU_ERR_INTERNAL; }
	c->q = d_queries; c->nq = nq; c->ef = ef; c->labels = d_labels; c->dists = d_dists; c->counts = d_counts; c->done = d_done;
	c->busy = 1;
	pthread_t th;
	pthread_attr_t at;
	pthread_attr_init(&at);
	pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
	if (pthread_create(&th, &at, flags_worker, c) != 0) { c->busy = 0; return HNSW_GPU_ERR_NOMEM; }
	pthread_attr_destroy(&at);
	return HNSW_GPU_OK;
}

int hnsw_gpu_ctx_idle(hnsw_gpu_ctx *c) { return __atomic_load_n(&c->busy, __ATOMIC_ACQUIRE) ? 0 : 1; }

void *hnsw_gpu_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void  hnsw_gpu_host_free(void *p) { free(p); }

int hnsw_gpu_dist_batch(dist_func_t func, const coord_t *q, const coord_t *rows, size_t nrows, size_t dim, dist_t *out)
{
	for (size_t i = 0; i < nrows; i++) out[i] = port_dist((int) func, q, rows + i * dim, dim);
	return HNSW_GPU_OK;
}

/* ---- row shards behind a front (server_main.cpp, HGS_OP_SHARD_*): the device-pointer search forms, the merge, and the exchange buffer
 * shared between PROCESSES.  "Device memory" of the double is host memory, so the shared buffer is a POSIX shared-memory object: its
 * name and size travel in the 64-byte handle. ---- */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static int dev_search(hnsw_gpu_index *ix, const coord_t *q, size_t nq, size_t ef, label_t *labels, dist_t *dists, uint32_t *counts)
{
	const size_t dim = ix->meta.dim;
	for (size_t i = 0; i < nq; i++)
	{
		size_t n = 0;
		for (size_t k = 0; k < ef; k++) { labels[i * ef + k] = ~(label_t) 0; if (dists) dists[i * ef + k] = 1.0f / 0.0f; }
		if (port_search(ix->p, q + i * dim, ef, labels + i * ef, dists ? dists + i * ef : NULL, &n, NULL, NULL) != 0) return HNSW_GPU_ERR_INTERNAL;
		if (counts) counts[i] = (uint32_t) n;
	}
	return HNSW_GPU_OK;
}

int hnsw_gpu_search_batch_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef, label_t *d_labels, dist_t *d_dists,
							  uint32_t *d_counts, uint32_t *d_stats, void *stream)
{
	(void) d_stats; (void) stream;
	return dev_search(ix, d_queries, nq, ef, d_labels, d_dists, d_counts);
}

int hnsw_gpu_search_batch_ctx(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef, label_t *d_labels, dist_t *d_dists,
							  uint32_t *d_counts, uint32_t *d_stats, void *stream)
{
	(void) d_stats; (void) stream;
	return dev_search(c->ix, d_queries, nq, ef, d_labels, d_dists, d_counts);
}

int hnsw_gpu_last_search_ms(hnsw_gpu_index *ix, float *ms) { (void) ix; *ms = 0.f; return HNSW_GPU_OK; }
int hnsw_gpu_device_wait(int device, void *stream) { (void) device; (void) stream; return HNSW_GPU_OK; }

/* the ef best by (distance, label) of nlists ascending lists per query (include/hnsw_gpu.h; csrc/gpu_sharded.hip, topk_merge_kernel) */
int hnsw_gpu_merge_topk_strided_dev(int device, const label_t *in_labels, size_t lstride, const dist_t *in_dists, size_t dstride, size_t nlists,
									size_t nq, size_t ef, label_t *out_labels, dist_t *out_dists, uint32_t *out_counts, void *stream)
{
	(void) device; (void) stream;
	size_t *at = (size_t *) malloc(nlists * sizeof(size_t));
	if (!at) return HNSW_GPU_ERR_NOMEM;
	for (size_t q = 0; q < nq; q++)
	{
		for (size_t l = 0; l < nlists; l++) at[l] = 0;
		uint32_t n = 0;
		for (size_t k = 0; k < ef; k++)
		{
			size_t best = nlists;
			for (size_t l = 0; l < nlists; l++)
			{
				if (at[l] >= ef) continue;
				const label_t la = in_labels[l * lstride + q * ef + at[l]];
				if (la == ~(label_t) 0) continue;                  /* the list's padded tail */
				const dist_t da = in_dists[l * dstride + q * ef + at[l]];
				if (best == nlists) { best = l; continue; }
				const dist_t db = in_dists[best * dstride + q * ef + at[best]];
				const label_t lb = in_labels[best * lstride + q * ef + at[best]];
				if (da < db || (da == db && la < lb)) best = l;
			}
			if (best == nlists) { out_labels[q * ef + k] = ~(label_t) 0; if (out_dists) out_dists[q * ef + k] = 1.0f / 0.0f; continue; }
			out_labels[q * ef + k] = in_labels[best * lstride + q * ef + at[best]];
			if (out_dists) out_dists[q * ef + k] = in_dists[best * dstride + q * ef + at[best]];
			at[best]++;
			n++;
		}
		out_counts[q] = n;
	}
	free(at);
	return HNSW_GPU_OK;
}

typedef struct { char name[40]; unsigned long long bytes; } DoubleHandle;      /* what the 64 bytes of an hnsw_gpu_ipc_handle hold here */
static int g_shared_seq = 0;
typedef struct SharedMap { void *p; size_t bytes; char name[40]; int owner; struct SharedMap *next; } SharedMap;
static SharedMap *g_shared = NULL;
static pthread_mutex_t g_shared_mu = PTHREAD_MUTEX_INITIALIZER;

static int shared_map(const char *name, size_t bytes, int create, void **out)
{
	const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
	if (fd < 0) { snprintf(t_err, sizeof(t_err), "double: shm_open(%s) failed", name); return HNSW_GPU_ERR_HIP; }
	if (create && ftruncate(fd, (off_t) bytes) != 0) { close(fd); shm_unlink(name); return HNSW_GPU_ERR_NOMEM; }
	void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) { if (create) shm_unlink(name); snprintf(t_err, sizeof(t_err), "double: mmap of %s failed", name); return HNSW_GPU_ERR_NOMEM; }
	SharedMap *m = (SharedMap *) calloc(1, sizeof(SharedMap));
	m->p = p; m->bytes = bytes; m->owner = create; snprintf(m->name, sizeof(m->name), "%s", name);
	pthread_mutex_lock(&g_shared_mu);
	m->next = g_shared; g_shared = m;
	pthread_mutex_unlock(&g_shared_mu);
	*out = p;
	return HNSW_GPU_OK;
}

static int shared_unmap(void *p, int owner)
{
	pthread_mutex_lock(&g_shared_mu);
	SharedMap **at = &g_shared;
	while (*at && (*at)->p != p) at = &(*at)->next;
	SharedMap *m = *at;
	if (m) *at = m->next;
	pthread_mutex_unlock(&g_shared_mu);
	if (!m) { snprintf(t_err, sizeof(t_err), "double: not a shared buffer"); return HNSW_GPU_ERR_ARG; }
	munmap(m->p, m->bytes);
	if (owner && m->owner) shm_unlink(m->name);
	free(m);
	return HNSW_GPU_OK;
}

int hnsw_gpu_shared_alloc(int device, size_t bytes, void **d_ptr, hnsw_gpu_ipc_handle *handle)
{
	(void) device;
	DoubleHandle h;
	memset(&h, 0, sizeof(h));
	snprintf(h.name, sizeof(h.name), "/hgsd_%d_%d", (int) getpid(), __atomic_add_fetch(&g_shared_seq, 1, __ATOMIC_RELAXED));
	h.bytes = bytes;
	const int rc = shared_map(h.name, bytes, 1, d_ptr);
	if (rc != HNSW_GPU_OK) return rc;
	memset(handle, 0, sizeof(*handle));
	memcpy(handle, &h, sizeof(h));
	return HNSW_GPU_OK;
}

int hnsw_gpu_shared_open(int device, const hnsw_gpu_ipc_handle *handle, void **d_ptr)
{
	(void) device;
	DoubleHandle h;
	memcpy(&h, handle, sizeof(h));
	h.name[sizeof(h.name) - 1] = 0;
	if (h.name[0] != '/' || h.bytes == 0) { snprintf(t_err, sizeof(t_err), "double: not a handle"); return HNSW_GPU_ERR_ARG; }
	return shared_map(h.name, (size_t) h.bytes, 0, d_ptr);
}

int hnsw_gpu_shared_close(int device, void *d_ptr) { (void) device; return shared_unmap(d_ptr, 0); }
int hnsw_gpu_shared_free(int device, void *d_ptr) { (void) device; return shared_unmap(d_ptr, 1); }
