/*
 * oracle/hnsw_port.c — TEST INFRASTRUCTURE ONLY: the CPU oracle.
 *
 * A plain-C restatement of the reference hot path (distfunc.c + hnswalg.cpp),
 * written from the algorithm, each function citing the reference lines it
 * follows.  It is the checker for the HIP path; nothing in the product imports,
 * links or executes it (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do).
 *
 * PINNING.  The restatement is pinned against (a) the orderings of the
 * reference's pg_regress expected files (tests/golden/knn_expected.json, restated
 * from test/expected/knn.out:16-19,39-42,56-59,97-122, gh-2.out:5-8, gh-3.out:9-14)
 * and (b) the UNMODIFIED reference sources compiled into oracle/_ref (see
 * oracle/Makefile) on seeded graphs: tests/test_oracle_golden.py; and, with the
 * reference's own distance function plugged in (port_set_dist_fn), the traversal
 * restated here returns the reference's result arrays EXACTLY, query for query
 * (tests/test_oracle_golden.py::test_port_with_reference_distances_is_the_reference).
 *
 * ONE DELIBERATE, DOCUMENTED DIFFERENCE: summation order.  The reference is built
 * with -Ofast (Makefile:14), so its float summation order is whatever the
 * compiler chose and is not reproducible bit-for-bit even
 * between two CPU builds.  The oracle instead fixes ONE canonical order — the
 * order the gfx950 kernels use — so that oracle and device agree BIT-EXACTLY and
 * every traversal decision (hnswalg.cpp:70,99) is identical:
 *
 *   - 64 strided partial sums: element e accumulates into s[e % 64], in
 *     increasing e, with a fused multiply-add (one rounding per element);
 *   - t[l] = (s[4l] + s[4l+1]) + (s[4l+2] + s[4l+3])        for l = 0..15;
 *   - xor-butterfly over the 16 t's in the order 1, 2, 4, 8:
 *         t[l] <- t[l] + t[l ^ off]   (all l simultaneously);   result = t[0];
 *   - epilogues as the reference writes them: sqrtf() for L2 (distfunc.c:64,129),
 *     double-precision 1 - dot/sqrt(na*nb) for cosine (distfunc.c:144), none for
 *     Manhattan (distfunc.c:154).
 *
 * Against the reference binary this differs by float round-off only
 * (<= 1e-5 relative, the north-star tolerance; measured ~1e-7).
 */
#include <math.h>
#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { PORT_L2 = 0, PORT_COSINE = 1, PORT_MANHATTAN = 2 };   /* embedding.h:22-26 */

/* ------------------------------------------------------------------------- */
/* Distances (distfunc.c) in the canonical summation order.                  */
/* ------------------------------------------------------------------------- */

static inline float canon_reduce64(const float *s)
{
	float t[16], u[16];
	for (int l = 0; l < 16; l++)
		t[l] = (s[4 * l] + s[4 * l + 1]) + (s[4 * l + 2] + s[4 * l + 3]);
	for (int off = 1; off < 16; off <<= 1)
	{
		for (int l = 0; l < 16; l++) u[l] = t[l] + t[l ^ off];
		for (int l = 0; l < 16; l++) t[l] = u[l];
	}
	return t[0];
}

/* distfunc.c:28-65 / 67-118 / 121-130: sqrtf(sum (x-y)^2). */
static float port_l2(const float *q, const float *x, size_t dim)
{
	float s[64];
	size_t e = 0;
	for (int j = 0; j < 64; j++) s[j] = 0.0f;
	for (; e + 64 <= dim; e += 64)
		for (int j = 0; j < 64; j++)
		{
			float d = q[e + j] - x[e + j];
			s[j] = __builtin_fmaf(d, d, s[j]);
		}
	for (int j = 0; e + j < dim; j++)
	{
		float d = q[e + j] - x[e + j];
		s[j] = __builtin_fmaf(d, d, s[j]);
	}
	return sqrtf(canon_reduce64(s));
}

/* distfunc.c:133-145: 1 - dot/sqrt(norma*normb); product in float, sqrt, divide
 * and subtract in double, result narrowed to float (distfunc.c:144). */
static float port_cosine(const float *q, const float *x, size_t dim)
{
	float sd[64], sa[64], sb[64];
	size_t e = 0;
	for (int j = 0; j < 64; j++) sd[j] = sa[j] = sb[j] = 0.0f;
	for (; e + 64 <= dim; e += 64)
		for (int j = 0; j < 64; j++)
		{
			float a = q[e + j], b = x[e + j];
			sd[j] = __builtin_fmaf(a, b, sd[j]);
			sa[j] = __builtin_fmaf(a, a, sa[j]);
			sb[j] = __builtin_fmaf(b, b, sb[j]);
		}
	for (int j = 0; e + j < dim; j++)
	{
		float a = q[e + j], b = x[e + j];
		sd[j] = __builtin_fmaf(a, b, sd[j]);
		sa[j] = __builtin_fmaf(a, a, sa[j]);
		sb[j] = __builtin_fmaf(b, b, sb[j]);
	}
	float dot = canon_reduce64(sd), na = canon_reduce64(sa), nb = canon_reduce64(sb);
	float prod = na * nb;
	double r = 1.0 - (double) dot / sqrt((double) prod);
	return (float) r;
}

/* distfunc.c:147-155: sum |x-y|. */
static float port_manhattan(const float *q, const float *x, size_t dim)
{
	float s[64];
	size_t e = 0;
	for (int j = 0; j < 64; j++) s[j] = 0.0f;
	for (; e + 64 <= dim; e += 64)
		for (int j = 0; j < 64; j++)
			s[j] = s[j] + fabsf(q[e + j] - x[e + j]);
	for (int j = 0; e + j < dim; j++)
		s[j] = s[j] + fabsf(q[e + j] - x[e + j]);
	return canon_reduce64(s);
}

/* distfunc.c:157-174: dispatch on dist_func_t. */
float port_dist(int func, const float *q, const float *x, size_t dim)
{
	switch (func)
	{
		case PORT_L2:        return port_l2(q, x, dim);
		case PORT_COSINE:    return port_cosine(q, x, dim);
		default:             return port_manhattan(q, x, dim);
	}
}

void port_dist_many(int func, const float *q, const float *rows, size_t nrows, size_t dim, float *out)
{
	for (size_t i = 0; i < nrows; i++)
		out[i] = port_dist(func, q, rows + i * dim, dim);
}

/* ------------------------------------------------------------------------- */
/* Index image: same bytes as the host's element array (embedding.c:222-228). */
/* ------------------------------------------------------------------------- */

typedef float (*port_dist_fn)(int func, const float *a, const float *b, size_t dim);

typedef struct PortIndex
{
	size_t dim, M, maxM, efc, efs;
	int    func;
	port_dist_fn dist;        /* port_dist (canonical order) unless a test plugs in the reference's hnsw_dist_func */
	port_dist_fn dist2;       /* NULL, or a second arithmetic that shadows every decision (see PortStats.div_*) */
	size_t off_data, off_label, elem_size;
	char  *data;
	size_t n, cap;
} PortIndex;

static inline uint32_t *el_links(const PortIndex *ix, uint32_t i)
{ return (uint32_t *) (ix->data + (size_t) i * ix->elem_size); }
static inline float *el_vec(const PortIndex *ix, uint32_t i)
{ return (float *) (ix->data + (size_t) i * ix->elem_size + ix->off_data); }
static inline uint64_t el_label(const PortIndex *ix, uint32_t i)
{ uint64_t l; memcpy(&l, ix->data + (size_t) i * ix->elem_size + ix->off_label, 8); return l; }

PortIndex *port_create(size_t dim, size_t M, size_t efc, size_t efs, int func, size_t capacity)
{
	PortIndex *ix = (PortIndex *) calloc(1, sizeof(PortIndex));
	if (!ix) return NULL;
	ix->dim = dim; ix->M = M; ix->maxM = 2 * M;           /* embedding.c:224 */
	ix->efc = efc; ix->efs = efs; ix->func = func;
	ix->dist = port_dist;
	ix->off_data = (ix->maxM + 1) * 4;                    /* embedding.c:226 */
	ix->off_label = ix->off_data + dim * 4;               /* embedding.c:227 */
	ix->elem_size = ix->off_label + 8;                    /* embedding.c:228 */
	ix->cap = capacity ? capacity : 16;
	ix->data = (char *) calloc(ix->cap, ix->elem_size);
	if (!ix->data) { free(ix); return NULL; }
	return ix;
}
/* Swap the distance function: signature of hnsw_dist_func (embedding.h:55).  With the reference's own
 * function (oracle/_ref) the restated traversal must reproduce the reference's results exactly, which
 * separates "is the traversal restated faithfully" from "what does the summation order change". */
void   port_set_dist_fn(PortIndex *ix, void *fn) { ix->dist = fn ? (port_dist_fn) fn : port_dist; }
/* Shadow arithmetic: every evaluation is also scored with `fn`; the walk follows ix->dist, and the first
 * decision that `fn`'s values would have taken differently is recorded per query (PortStats.div_*). */
void   port_set_dist2_fn(PortIndex *ix, void *fn) { ix->dist2 = (port_dist_fn) fn; }
void   port_destroy(PortIndex *ix) { if (ix) { free(ix->data); free(ix); } }
size_t port_count(PortIndex *ix) { return ix->n; }
void  *port_data(PortIndex *ix) { return ix->data; }
size_t port_elem_size(PortIndex *ix) { return ix->elem_size; }

static int port_reserve(PortIndex *ix, size_t want)
{
	if (want <= ix->cap) return 0;
	size_t ncap = ix->cap * 2 < want ? want : ix->cap * 2;
	char *nd = (char *) realloc(ix->data, ncap * ix->elem_size);
	if (!nd) return -1;
	memset(nd + ix->cap * ix->elem_size, 0, (ncap - ix->cap) * ix->elem_size);
	ix->data = nd; ix->cap = ncap;
	return 0;
}

int port_load_raw(PortIndex *ix, const void *bytes, size_t n)
{
	if (port_reserve(ix, n) != 0) return -1;
	memcpy(ix->data, bytes, n * ix->elem_size);
	ix->n = n;
	return 0;
}

void port_set_deleted(PortIndex *ix, uint32_t idx, int deleted)
{
	char *p = ix->data + (size_t) idx * ix->elem_size + ix->off_label;
	uint64_t l; memcpy(&l, p, 8);
	if (deleted) l |= (uint64_t) 1 << 48; else l &= ~((uint64_t) 1 << 48);   /* embedding.c:920-926 */
	memcpy(p, &l, 8);
}

/* ------------------------------------------------------------------------- */
/* std::priority_queue<std::pair<float, K>> restated: a binary max-heap under */
/* the lexicographic pair order (first, then second).  Only the extremes of a */
/* strict total order are observable, so any heap shape gives the reference's */
/* results; NaN keys are outside the contract (SURVEY.md §7 "Cosine numerics").*/
/* ------------------------------------------------------------------------- */

typedef struct { float d; uint64_t k; } HPair;
typedef struct { HPair *a; size_t n, cap; } Heap;

static inline bool pair_less(HPair x, HPair y)
{ return x.d < y.d || (!(y.d < x.d) && x.k < y.k); }

static void heap_init(Heap *h) { h->a = NULL; h->n = h->cap = 0; }
static void heap_free(Heap *h) { free(h->a); h->a = NULL; h->n = h->cap = 0; }
static void heap_push(Heap *h, float d, uint64_t k)
{
	if (h->n == h->cap)
	{
		h->cap = h->cap ? h->cap * 2 : 64;
		h->a = (HPair *) realloc(h->a, h->cap * sizeof(HPair));
		if (!h->a) { fprintf(stderr, "hnsw_port: out of memory\n"); abort(); }
	}
	size_t i = h->n++;
	HPair v = { d, k };
	while (i > 0)
	{
		size_t p = (i - 1) / 2;
		if (!pair_less(h->a[p], v)) break;
		h->a[i] = h->a[p];
		i = p;
	}
	h->a[i] = v;
}
static inline HPair heap_top(const Heap *h) { return h->a[0]; }
static void heap_pop(Heap *h)
{
	HPair v = h->a[--h->n];
	size_t i = 0, n = h->n;
	for (;;)
	{
		size_t c = 2 * i + 1;
		if (c >= n) break;
		if (c + 1 < n && pair_less(h->a[c], h->a[c + 1])) c++;
		if (!pair_less(v, h->a[c])) break;
		h->a[i] = h->a[c];
		i = c;
	}
	if (n) h->a[i] = v;
}

/* ------------------------------------------------------------------------- */
/* searchBaseLayer — hnswalg.cpp:42-114.                                     */
/* ------------------------------------------------------------------------- */

/* min_margin: the smallest relative gap |a-b| / max(|a|,|b|,1e-6) over every comparison of two DIFFERENT
 * elements' distances that steers the walk or the output — the decision-margin instrumentation the
 * parity tests use to classify a mismatch against the reference binary: two implementations whose
 * distances agree to delta (relative) can only take different decisions where a margin is <= 2*delta. */
/* div_kind / div_margin (shadow arithmetic, port_set_dist2_fn): the FIRST decision of the walk that the second
 * arithmetic would have taken differently — 0 none, 1 stop test (:70), 2 accept test (:99), 3 which candidate is
 * popped (:69,73), 4 which result is evicted (:105), 5 output order (:236,246) — and the relative gap, in the
 * walk's own arithmetic, between the two values that decision compared.  Up to that decision both arithmetics
 * walk identically (same heaps, element for element), so: div_kind == 0  =>  the second arithmetic returns the
 * same ids; and a query whose ids differ has div_kind != 0, with div_margin = how close the call was. */
typedef struct { uint64_t evals, hops; float min_margin; int div_kind; float div_margin;
                 uint32_t *pops; size_t pops_cap;      /* optional: the pop sequence (hnswalg.cpp:73), first pops_cap of them */
               } PortStats;
#define PORT_STATS_INIT { 0, 0, INFINITY, 0, INFINITY, NULL, 0 }

static inline float rel_gap(float a, float b)
{
	return fabsf(a - b) / fmaxf(fmaxf(fabsf(a), fabsf(b)), 1e-6f);
}
static inline void note_div(PortStats *st, int kind, float a, float b)
{
	if (st->div_kind == 0) { st->div_kind = kind; st->div_margin = rel_gap(a, b); }
}

static inline void note_margin(PortStats *st, float a, float b)
{
	float m = rel_gap(a, b);
	if (m < st->min_margin) st->min_margin = m;
}

/* the runner-up of a heap (the larger child of the root under the pair order); h->n >= 2 */
static inline HPair heap_second(const Heap *h)
{
	if (h->n >= 3 && pair_less(h->a[1], h->a[2])) return h->a[2];
	return h->a[1];
}

/* Leaves the <= ef nearest visited elements in `top` (max-heap on (dist, idx)).
 * d2: NULL, or an array of ix->n floats (contents irrelevant) for the shadow arithmetic's distances. */
static void port_search_base_layer(const PortIndex *ix, const float *q, size_t ef, Heap *top, PortStats *st, float *d2)
{
	Heap cand;                          /* keys are (-dist, idx): hnswalg.cpp:53,63 */
	uint32_t ep = 0;                    /* enterpoint_node, embedding.c:235 */
	heap_init(top);
	if (ix->n == 0)                     /* hnsw_begin_read(entry) == false: hnswalg.cpp:56-57 */
		return;
	heap_init(&cand);
	size_t words = (ix->n + 31) / 32;
	uint32_t *visited = (uint32_t *) calloc(words, 4);     /* hnswalg.cpp:45-50 */
	if (!ix->dist2) d2 = NULL;

	float dist = ix->dist(ix->func, q, el_vec(ix, ep), ix->dim);    /* :59 */
	if (d2) d2[ep] = ix->dist2(ix->func, q, el_vec(ix, ep), ix->dim);
	st->evals++;
	heap_push(top, dist, ep);                                       /* :62 */
	heap_push(&cand, -dist, ep);                                    /* :63 */
	visited[ep >> 5] = 1u << (ep & 31);                             /* :64 */
	float lowerBound = dist;                                        /* :65 */

	while (cand.n)                                                  /* :67 */
	{
		HPair cur = heap_top(&cand);
		if (d2)                                                     /* would the shadow pop another candidate? */
		{
			HPair mine = { -d2[cur.k], cur.k };
			for (size_t i = 1; i < cand.n; i++)
			{
				HPair o = { -d2[cand.a[i].k], cand.a[i].k };
				if (pair_less(mine, o)) { note_div(st, 3, cur.d, cand.a[i].d); break; }
			}
			if ((-cur.d > lowerBound) != (d2[cur.k] > d2[heap_top(top).k]))
				note_div(st, 1, -cur.d, lowerBound);
		}
		if (cur.k != heap_top(top).k)                               /* margin of the stop test */
			note_margin(st, -cur.d, lowerBound);
		if (-cur.d > lowerBound)                                    /* :70-71 */
			break;
		if (cand.n >= 2)                                            /* margin of "which candidate is next" */
			note_margin(st, cur.d, heap_second(&cand).d);
		heap_pop(&cand);                                            /* :73 */
		const uint32_t *links = el_links(ix, (uint32_t) cur.k);     /* :76 */
		size_t size = links[0];                                     /* :77 */
		if (st->pops && st->hops < st->pops_cap) st->pops[st->hops] = (uint32_t) cur.k;
		st->hops++;
		/* pass 1 (:79-88) only prefetches; pass 2 (:89-110): */
		for (size_t j = 0; j < size; j++)
		{
			uint32_t t = links[1 + j];
			if (visited[t >> 5] & (1u << (t & 31)))
				continue;
			visited[t >> 5] |= 1u << (t & 31);                      /* :93 */
			dist = ix->dist(ix->func, q, el_vec(ix, t), ix->dim);   /* :95-97 */
			if (d2) d2[t] = ix->dist2(ix->func, q, el_vec(ix, t), ix->dim);
			st->evals++;
			if (top->n >= ef)                                       /* margin of the accept test */
			{
				note_margin(st, heap_top(top).d, dist);
				if (d2 && (heap_top(top).d > dist) != (d2[heap_top(top).k] > d2[t]))
					note_div(st, 2, heap_top(top).d, dist);
			}
			if (heap_top(top).d > dist || top->n < ef)              /* :99 */
			{
				heap_push(&cand, -dist, t);                         /* :100 */
				heap_push(top, dist, t);                            /* :102 */
				if (top->n > ef)                                    /* :104-105 */
				{
					if (top->n >= 2)                                /* margin of "which result is evicted" */
						note_margin(st, heap_top(top).d, heap_second(top).d);
					if (d2)                                         /* would the shadow evict another result? */
					{
						HPair w = heap_top(top), mine = { d2[w.k], w.k };
						for (size_t i = 1; i < top->n; i++)
						{
							HPair o = { d2[top->a[i].k], top->a[i].k };
							if (pair_less(mine, o)) { note_div(st, 4, w.d, top->a[i].d); break; }
						}
					}
					heap_pop(top);
				}
				lowerBound = heap_top(top).d;                       /* :107 */
			}
		}
	}
	free(visited);
	heap_free(&cand);
}

/* Base-layer result as arrays ascending by (dist, idx) — used by device parity tests. */
int port_search_base(PortIndex *ix, const float *q, size_t ef, uint32_t *idx_out, float *dist_out,
					 size_t *n_out, uint32_t *evals, uint32_t *hops)
{
	Heap top;
	PortStats st = PORT_STATS_INIT;
	port_search_base_layer(ix, q, ef, &top, &st, NULL);
	size_t n = top.n;
	for (size_t i = n; i-- != 0;)
	{
		HPair p = heap_top(&top);
		idx_out[i] = (uint32_t) p.k;
		dist_out[i] = p.d;
		heap_pop(&top);
	}
	heap_free(&top);
	*n_out = n;
	if (evals) *evals = (uint32_t) st.evals;
	if (hops)  *hops = (uint32_t) st.hops;
	return 0;
}

/* ------------------------------------------------------------------------- */
/* searchKnn + hnsw_search — hnswalg.cpp:234-252, 256-277.                   */
/* ------------------------------------------------------------------------- */

/* out arrays must hold ef entries.  Result ascending by (dist, label), vacuumed
 * labels dropped (hnswalg.cpp:245).  dist_out may be NULL (the reference does
 * not return distances; the device batch API does). */
static int port_search_m(PortIndex *ix, const float *q, size_t ef, uint64_t *label_out, float *dist_out,
						 size_t *n_out, uint32_t *evals, uint32_t *hops, PortStats *st_out, float *d2)
{
	Heap top, res;
	PortStats st = PORT_STATS_INIT;
	if (st_out) { st.pops = st_out->pops; st.pops_cap = st_out->pops_cap; }   /* (a caller that wants the pop sequence) */
	if (!ix->dist2) d2 = NULL;
	port_search_base_layer(ix, q, ef, &top, &st, d2);        /* :237 */
	while (top.n > ef) heap_pop(&top);                       /* :238-240 */
	heap_init(&res);
	/* shadow distance of each returned label (labels of live elements are unique) */
	size_t nsh = 0;
	uint64_t *sh_label = d2 ? (uint64_t *) malloc((top.n ? top.n : 1) * 8) : NULL;
	float    *sh_d2    = d2 ? (float *) malloc((top.n ? top.n : 1) * 4) : NULL;
	while (top.n)                                            /* :241-249 */
	{
		HPair r = heap_top(&top);
		uint64_t label = el_label(ix, (uint32_t) r.k);
		if (!((label >> 48) & 1))                            /* hnsw_is_deleted, embedding.c:948-953 */
		{
			heap_push(&res, r.d, label);
			if (d2) { sh_label[nsh] = label; sh_d2[nsh] = d2[r.k]; nsh++; }
		}
		heap_pop(&top);
	}
	size_t n = res.n;
	HPair prev = { 0.f, 0 }, prev2 = { 0.f, 0 };
	for (size_t i = n; i-- != 0;)                            /* back-to-front fill, :265-269 */
	{
		HPair p = heap_top(&res);
		label_out[i] = p.k;
		if (dist_out) dist_out[i] = p.d;
		HPair p2 = { 0.f, p.k };
		if (d2)
			for (size_t j = 0; j < nsh; j++)
				if (sh_label[j] == p.k) { p2.d = sh_d2[j]; break; }
		if (i + 1 < n)
		{
			note_margin(&st, p.d, prev.d);                   /* margin of the output order */
			if (d2 && pair_less(prev2, p2)) note_div(&st, 5, p.d, prev.d);
		}
		prev = p; prev2 = p2;
		heap_pop(&res);
	}
	free(sh_label); free(sh_d2);
	heap_free(&top);
	heap_free(&res);
	*n_out = n;
	if (evals) *evals = (uint32_t) st.evals;
	if (hops)  *hops = (uint32_t) st.hops;
	if (st_out) *st_out = st;
	return 0;
}

/* hnsw_search plus the walk's pop sequence (what hnsw_gpu_search_trace returns on the device); base != 0: searchBaseLayer
 * only, element numbers in label_out. */
int port_search_trace(PortIndex *ix, const float *q, size_t ef, int base, uint64_t *label_out, float *dist_out,
					  size_t *n_out, uint32_t *evals, uint32_t *pops, size_t pops_cap, uint32_t *npops)
{
	PortStats st = PORT_STATS_INIT;
	st.pops = pops; st.pops_cap = pops_cap;
	if (base)
	{
		Heap top;
		port_search_base_layer(ix, q, ef, &top, &st, NULL);
		size_t n = top.n;
		for (size_t i = n; i-- != 0;)
		{
			HPair p = heap_top(&top);
			label_out[i] = p.k;
			if (dist_out) dist_out[i] = p.d;
			heap_pop(&top);
		}
		heap_free(&top);
		*n_out = n;
		if (evals) *evals = (uint32_t) st.evals;
		*npops = (uint32_t) st.hops;
		return 0;
	}
	uint32_t ev = 0, hp = 0;
	PortStats out = PORT_STATS_INIT;
	out.pops = pops; out.pops_cap = pops_cap;
	int rc = port_search_m(ix, q, ef, label_out, dist_out, n_out, &ev, &hp, &out, NULL);
	if (evals) *evals = ev;
	*npops = hp;
	return rc;
}

int port_search(PortIndex *ix, const float *q, size_t ef, uint64_t *label_out, float *dist_out,
				size_t *n_out, uint32_t *evals, uint32_t *hops)
{
	return port_search_m(ix, q, ef, label_out, dist_out, n_out, evals, hops, NULL, NULL);
}

/* ------------------------------------------------------------------------- */
/* Insert path — hnswalg.cpp:117-232.                                        */
/* ------------------------------------------------------------------------- */

/* getNeighborsByHeuristic, hnswalg.cpp:117-153.  `top` is a max-heap on
 * (dist, idx); on return it holds at most NN survivors.
 * sh: NULL, or the shadow arithmetic's distance of every element of `top` to the same centre, indexed by element
 * number (port_set_dist2_fn): the first decision it would take differently is recorded in st->div_* — 6 which
 * candidate is considered next (:130-134), 7 the pair test (:144), [8 = the order of a written list, in
 * port_mutually_connect]. */
static void port_neighbors_by_heuristic(const PortIndex *ix, Heap *top, size_t NN, PortStats *st, const float *sh)
{
	if (top->n < NN)                                         /* :119-120 */
		return;
	Heap rs;                                                 /* (-dist, idx): closest first */
	heap_init(&rs);
	HPair *ret = (HPair *) malloc((NN ? NN : 1) * sizeof(HPair));
	size_t nret = 0;
	while (top->n)                                           /* :125-128 */
	{
		HPair p = heap_top(top);
		heap_push(&rs, -p.d, p.k);
		heap_pop(top);
	}
	while (rs.n)                                             /* :130 */
	{
		if (nret >= NN)                                      /* :131-132 */
			break;
		HPair cur = heap_top(&rs);
		if (sh)                                              /* would the shadow look at another candidate now? */
		{
			HPair mine = { -sh[cur.k], cur.k };
			for (size_t i = 1; i < rs.n; i++)
			{
				HPair o = { -sh[rs.a[i].k], rs.a[i].k };
				if (pair_less(mine, o)) { note_div(st, 6, cur.d, rs.a[i].d); break; }
			}
		}
		float dist_to_query = -cur.d;
		heap_pop(&rs);
		bool good = true;
		for (size_t i = 0; i < nret; i++)                    /* :137-148 */
		{
			float curdist = ix->dist(ix->func, el_vec(ix, (uint32_t) ret[i].k),
									  el_vec(ix, (uint32_t) cur.k), ix->dim);
			st->evals++;
			if (sh)
			{
				float curdist2 = ix->dist2(ix->func, el_vec(ix, (uint32_t) ret[i].k), el_vec(ix, (uint32_t) cur.k), ix->dim);
				if ((curdist < dist_to_query) != (curdist2 < sh[cur.k])) note_div(st, 7, curdist, dist_to_query);
			}
			if (curdist < dist_to_query) { good = false; break; }
		}
		if (good) ret[nret++] = cur;                         /* :149 */
	}
	for (size_t i = 0; i < nret; i++)                        /* :151-152 */
		heap_push(top, -ret[i].d, ret[i].k);
	free(ret);
	heap_free(&rs);
}

/* pops `h` (max-heap on (dist, idx)) into out[] farthest first, as :164-167 and :214-219 do; with a shadow
 * arithmetic, notes where ITS (dist, idx) order of the same elements would differ (kind 8) */
static size_t drain_farthest_first(Heap *h, uint32_t *out, PortStats *st, const float *sh)
{
	size_t n = 0;
	HPair prev = { 0, 0 };
	while (h->n)
	{
		HPair p = heap_top(h);
		if (sh && n > 0)
		{
			HPair p2 = { sh[p.k], p.k }, prev2 = { sh[prev.k], prev.k };
			if (pair_less(prev2, p2)) note_div(st, 8, p.d, prev.d);
		}
		out[n++] = (uint32_t) p.k;
		prev = p;
		heap_pop(h);
	}
	return n;
}

/* mutuallyConnectNewElement, hnswalg.cpp:155-223.  Returns 0, or -1 where the
 * reference throws.  d2q / d2n: NULL, or two arrays of ix->n floats for the shadow arithmetic — d2q holds its
 * distances to the new point for every element the walk scored, d2n is scratch for the re-selections. */
static int port_mutually_connect(PortIndex *ix, uint32_t cur_c, Heap *top, PortStats *st, const float *d2q, float *d2n)
{
	port_neighbors_by_heuristic(ix, top, ix->M, st, d2q);     /* :158 */
	uint32_t *res = (uint32_t *) malloc((top->n ? top->n : 1) * 4);
	size_t nres = drain_farthest_first(top, res, st, d2q);    /* :164-167: farthest first */

	uint32_t *mine = el_links(ix, cur_c);                     /* :169-181 */
	if (mine[0]) { free(res); return -1; }                    /* "Should be blank" */
	mine[0] = (uint32_t) nres;
	for (size_t i = 0; i < nres; i++)
	{
		if (mine[1 + i]) { free(res); return -1; }
		mine[1 + i] = res[i];
	}
	for (size_t i = 0; i < nres; i++)                         /* :183-222 */
	{
		if (res[i] == cur_c) { free(res); return -1; }        /* "Connection to the same element" */
		uint32_t *other = el_links(ix, res[i]);
		uint32_t sz = other[0];
		if (sz > ix->maxM) { free(res); return -1; }          /* "Bad sz_link_list_other" */
		if (sz < ix->maxM)                                    /* :194-196 */
		{
			other[1 + sz] = cur_c;
			other[0] = sz + 1;
		}
		else                                                  /* :197-220 */
		{
			const float *pc = el_vec(ix, res[i]);
			Heap cands;
			heap_init(&cands);
			float d_max = ix->dist(ix->func, el_vec(ix, cur_c), pc, ix->dim);    /* :200 */
			if (d2n) d2n[cur_c] = ix->dist2(ix->func, el_vec(ix, cur_c), pc, ix->dim);
			st->evals++;
			heap_push(&cands, d_max, cur_c);                                      /* :204 */
			for (uint32_t j = 0; j < sz; j++)                                     /* :206-211 */
			{
				uint32_t o = other[1 + j];
				heap_push(&cands, ix->dist(ix->func, el_vec(ix, o), pc, ix->dim), o);
				if (d2n) d2n[o] = ix->dist2(ix->func, el_vec(ix, o), pc, ix->dim);
				st->evals++;
			}
			port_neighbors_by_heuristic(ix, &cands, ix->maxM, st, d2n);           /* :212 */
			other[0] = (uint32_t) drain_farthest_first(&cands, other + 1, st, d2n);   /* :214-219 */
			heap_free(&cands);
		}
	}
	free(res);
	return 0;
}

/* bindPoint / hnsw_bind_point, hnswalg.cpp:225-232, 279-291.  st_out: NULL, or where the insert's decision record goes
 * (div_kind / div_margin under a shadow arithmetic: the first decision of the walk, the selections or the list orders that
 * the second arithmetic would take differently — none: it builds the same bytes). */
static int port_bind_point_st(PortIndex *ix, const float *point, uint32_t cur_c, PortStats *st_out)
{
	if (cur_c == 0)                                           /* :228 */
		return 0;
	Heap top;
	PortStats st = PORT_STATS_INIT;
	float *d2q = NULL, *d2n = NULL;
	if (ix->dist2)
	{
		d2q = (float *) malloc(ix->n * 4);
		d2n = (float *) malloc(ix->n * 4);
		if (!d2q || !d2n) { free(d2q); free(d2n); return -1; }
	}
	port_search_base_layer(ix, point, ix->efc, &top, &st, d2q);     /* :229 */
	int rc = port_mutually_connect(ix, cur_c, &top, &st, d2q, d2n);  /* :230 */
	heap_free(&top);
	free(d2q); free(d2n);
	if (st_out) *st_out = st;
	return rc;
}

int port_bind_point(PortIndex *ix, const float *point, uint32_t cur_c)
{
	return port_bind_point_st(ix, point, cur_c, NULL);
}

/* hnsw_add_point minus paging (embedding.c:606-701): store zero-linked, then bind. */
static long port_add_st(PortIndex *ix, const float *vec, uint64_t label, PortStats *st_out)
{
	if (port_reserve(ix, ix->n + 1) != 0) return -1;
	uint32_t idx = (uint32_t) ix->n;
	char *p = ix->data + (size_t) idx * ix->elem_size;
	memset(p, 0, ix->off_data);
	memcpy(p + ix->off_data, vec, ix->dim * 4);
	memcpy(p + ix->off_label, &label, 8);
	ix->n++;
	if (port_bind_point_st(ix, vec, idx, st_out) != 0) return -2;
	return (long) idx;
}

long port_add(PortIndex *ix, const float *vec, uint64_t label)
{
	return port_add_st(ix, vec, label, NULL);
}

/* One insert with its decision record under the shadow arithmetic (port_set_dist2_fn): div_kind 0 = the second arithmetic
 * builds the same lists from the same state; else the kind of the first decision it takes differently (1-4 the walk,
 * 6 / 7 getNeighborsByHeuristic, 8 the order of a written list) and the relative gap of the two values compared. */
long port_add_shadowed(PortIndex *ix, const float *vec, uint64_t label, int32_t *div_kind, float *div_margin)
{
	PortStats st = PORT_STATS_INIT;
	long r = port_add_st(ix, vec, label, &st);
	if (div_kind) *div_kind = st.div_kind;
	if (div_margin) *div_margin = st.div_margin;
	return r;
}

long port_add_many(PortIndex *ix, const float *vecs, const uint64_t *labels, size_t n)
{
	for (size_t i = 0; i < n; i++)
	{
		long r = port_add(ix, vecs + i * ix->dim, labels ? labels[i] : (uint64_t) ix->n);
		if (r < 0) return r;
	}
	return (long) ix->n;
}

/* ------------------------------------------------------------------------- */
/* Timed multi-query driver (cpu_baseline kind "port").                      */
/* ------------------------------------------------------------------------- */

typedef struct
{
	PortIndex *ix; const float *Q; size_t q0, q1, ef;
	uint64_t *labels; float *dists; uint32_t *counts, *evals, *hops; float *margins;
	int32_t *div_kind; float *div_margin;
} PortJob;

static void *port_worker(void *arg)
{
	PortJob *j = (PortJob *) arg;
	uint64_t *lab = (uint64_t *) malloc(j->ef * 8);
	float *dst = (float *) malloc(j->ef * 4);
	float *d2 = j->ix->dist2 ? (float *) malloc((j->ix->n ? j->ix->n : 1) * 4) : NULL;
	for (size_t q = j->q0; q < j->q1; q++)
	{
		size_t n; uint32_t ev, hp; PortStats st = PORT_STATS_INIT;
		port_search_m(j->ix, j->Q + q * j->ix->dim, j->ef, lab, dst, &n, &ev, &hp, &st, d2);
		if (j->margins) j->margins[q] = st.min_margin;
		if (j->div_kind) j->div_kind[q] = st.div_kind;
		if (j->div_margin) j->div_margin[q] = st.div_margin;
		if (j->labels) memcpy(j->labels + q * j->ef, lab, n * 8);
		if (j->dists)  memcpy(j->dists + q * j->ef, dst, n * 4);
		if (j->counts) j->counts[q] = (uint32_t) n;
		if (j->evals)  j->evals[q] = ev;
		if (j->hops)   j->hops[q] = hp;
	}
	free(lab); free(dst); free(d2);
	return NULL;
}

/* margins: NULL, or per query the smallest decision margin of its walk (PortStats.min_margin);
 * div_kind / div_margin: NULL, or per query the first decision the shadow arithmetic takes differently. */
double port_search_many_m(PortIndex *ix, const float *Q, size_t nq, size_t ef, int nthreads,
						  uint64_t *labels, float *dists, uint32_t *counts, uint32_t *evals, uint32_t *hops,
						  float *margins, int32_t *div_kind, float *div_margin);

double port_search_many(PortIndex *ix, const float *Q, size_t nq, size_t ef, int nthreads,
						uint64_t *labels, float *dists, uint32_t *counts, uint32_t *evals, uint32_t *hops)
{
	return port_search_many_m(ix, Q, nq, ef, nthreads, labels, dists, counts, evals, hops, NULL, NULL, NULL);
}

double port_search_many_m(PortIndex *ix, const float *Q, size_t nq, size_t ef, int nthreads,
						  uint64_t *labels, float *dists, uint32_t *counts, uint32_t *evals, uint32_t *hops,
						  float *margins, int32_t *div_kind, float *div_margin)
{
	if (nthreads < 1) nthreads = 1;
	if ((size_t) nthreads > nq && nq > 0) nthreads = (int) nq;
	PortJob *jobs = (PortJob *) calloc((size_t) nthreads, sizeof(PortJob));
	pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int t = 0; t < nthreads; t++)
	{
		PortJob *j = &jobs[t];
		j->ix = ix; j->Q = Q; j->ef = ef;
		j->q0 = nq * (size_t) t / (size_t) nthreads;
		j->q1 = nq * (size_t) (t + 1) / (size_t) nthreads;
		j->labels = labels; j->dists = dists; j->counts = counts; j->evals = evals; j->hops = hops; j->margins = margins; j->div_kind = div_kind; j->div_margin = div_margin;
		if (nthreads == 1) port_worker(j); else pthread_create(&th[t], NULL, port_worker, j);
	}
	for (int t = 0; t < nthreads && nthreads > 1; t++) pthread_join(th[t], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(jobs); free(th);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
