/*
 * tests/dropin_c/dist_bench.c — cost of ONE hnsw_dist_func call the way the SQL operators make it
 * (calc_distance, embedding.c:1022-1046: one pair per fmgr call), against any library that exports the
 * symbol: libembedding_gpu.so / libembedding_gpuc.so (product) or oracle/_ref/libpgemb_ref.so (reference).
 *
 *   usage: dist_bench <library.so> <dim> <calls>      prints "<func> <ns per call> <checksum>" per metric
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef float (*dist_fn)(int, const float *, const float *, size_t);

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	void *h = dlopen(argv[1], RTLD_LAZY | RTLD_GLOBAL);
	if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
	dist_fn f = (dist_fn) dlsym(h, "hnsw_dist_func");
	void (*init)(void) = (void (*)(void)) dlsym(h, "hnsw_init_dist_func");
	if (!f || !init) { fprintf(stderr, "symbols missing\n"); return 1; }
	init();
	const size_t dim = (size_t) atol(argv[2]);
	const long calls = atol(argv[3]);
	const int rows = 512;
	float *q = malloc(dim * 4), *x = malloc(dim * rows * 4);
	unsigned long long s = 88172645463325252ull;
	for (size_t i = 0; i < dim; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; q[i] = (float) (s >> 40) / 16777216.f; }
	for (size_t i = 0; i < dim * rows; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; x[i] = (float) (s >> 40) / 16777216.f; }
	for (int func = 0; func < 3; func++)
	{
		double sum = 0;
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		for (long c = 0; c < calls; c++) sum += f(func, q, x + (size_t) (c % rows) * dim, dim);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		const double ns = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / (double) calls;
		printf("%d %.1f %.6f\n", func, ns, sum);
	}
	return 0;
}
