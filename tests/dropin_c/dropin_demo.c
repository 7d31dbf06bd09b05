/*
 * tests/dropin_c/dropin_demo.c — the drop-in seen from C, the way embedding.c uses it.
 *
 * This translation unit plays the part of the Postgres glue: it owns the storage (through
 * oracle/flat_host.c, compiled next to it, which supplies hnsw_begin_read & co) and calls ONLY the
 * four symbols of embedding.h:46-47,55-56.  It is linked against libembedding_gpu.so where the
 * reference links hnswalg.o + distfunc.o — nothing else changes.
 *
 *   usage: dropin_demo <n> <dim> <m> <efc> <efs> <nq> [key]
 * With a key — and only when linked against the server client library, libembedding_gpuc.so — the
 * (still empty) index is first attached to the server-side mirror of that key, the way
 * hnsw_beginscan / hnsw_insert would (hnsw_gpu_server.h): inserts then extend that one mirror.
 * Rows/queries come from a fixed LCG; output: for every query the labels hnsw_search() returns.
 */
#include <stdio.h>
#include <stdlib.h>
#include "hnsw_abi.h"

/* flat_host.c */
typedef struct FlatIndex FlatIndex;
FlatIndex *flat_create(size_t dim, size_t M, size_t efc, size_t efs, int dist_func, size_t capacity);
long flat_add(FlatIndex *f, const coord_t *vec, label_t label);
HnswMetadata *flat_meta(FlatIndex *f);

/* libembedding_gpuc.so only (weak: absent in the other two link arrangements) */
extern int hnsw_gpu_remote_attach(HnswMetadata *meta, uint64_t key, uint64_t generation) __attribute__((weak));

static unsigned long long lcg = 88172645463325252ull;
static float rnd(void)
{
	lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
	return (float) ((lcg >> 40) & 0xFFFF) / 65536.0f;          /* [0,1) with 16 bits: exact in fp32 */
}

int main(int argc, char **argv)
{
	if (argc < 7) return 2;
	size_t n = (size_t) atol(argv[1]), dim = (size_t) atol(argv[2]), m = (size_t) atol(argv[3]);
	size_t efc = (size_t) atol(argv[4]), efs = (size_t) atol(argv[5]), nq = (size_t) atol(argv[6]);
	hnsw_init_dist_func();                                  /* _PG_init, embedding.c:150 */
	FlatIndex *f = flat_create(dim, m, efc, efs, DIST_L2, n);
	float *v = (float *) malloc(dim * sizeof(float));
	if (argc > 7 && hnsw_gpu_remote_attach && hnsw_gpu_remote_attach(flat_meta(f), strtoull(argv[7], NULL, 0), 1) != 0)
	{
		fprintf(stderr, "attach failed\n");
		return 1;
	}
	for (size_t i = 0; i < n; i++)
	{
		for (size_t d = 0; d < dim; d++) v[d] = rnd() + (float) (i % 7);
		if (flat_add(f, v, (label_t) (1000 + i)) < 0)       /* store + hnsw_bind_point, embedding.c:606-701 */
		{
			fprintf(stderr, "insert %zu failed\n", i);
			return 1;
		}
	}
	HnswMetadata *meta = flat_meta(f);
	for (size_t q = 0; q < nq; q++)
	{
		for (size_t d = 0; d < dim; d++) v[d] = rnd() + (float) (q % 7);
		size_t nres = 0;
		label_t *res = NULL;
		if (!hnsw_search(meta, v, &nres, &res))             /* embedding.c:317 */
		{
			fprintf(stderr, "HNSW index search failed\n");
			return 1;
		}
		printf("q%zu:", q);
		for (size_t i = 0; i < nres; i++) printf(" %llu", (unsigned long long) res[i]);
		printf(" | d0=%.6f\n", (double) hnsw_dist_func(DIST_L2, v, v, dim));
		free(res);                                          /* embedding.c:327 */
	}
	return 0;
}
