/*
 * tests/dropin_c/server_clients.c — many Postgres-like backends against one hnsw_gpu_server.
 *
 * Forks <nproc> single-threaded processes.  Each one is a miniature backend: it owns an
 * HnswMetadata (through oracle/flat_host.c, which also supplies the storage callbacks the library
 * imports), attaches it to the server-side mirror (key, generation) the way hnsw_beginscan would,
 * and then calls the reference's own entry point hnsw_search() (embedding.h:46), one query per call
 * (embedding.c:317), for its share of the query file.  The labels every call returns go to a shared
 * output file, so the caller can compare them with the oracle; the parent times the whole run.
 *
 *   usage: server_clients <key> <gen> <dim> <m> <efc> <efs> <func> <queries.f32> <nq> <nproc> <out.u64> <rounds>
 *   env:   PG_EMBEDDING_GPU_SERVER = socket path
 *   out:   nq*efs labels (unused tail ~0) followed by nq u64 counts
 *   stdout: one JSON line {"nproc":..,"nq":..,"rounds":..,"seconds":..,"qps":..}
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "hnsw_gpu_server.h"

typedef struct FlatIndex FlatIndex;
FlatIndex *flat_create(size_t dim, size_t M, size_t efc, size_t efs, int dist_func, size_t capacity);
HnswMetadata *flat_meta(FlatIndex *f);

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

int main(int argc, char **argv)
{
	if (argc < 13) { fprintf(stderr, "usage: see the file header\n"); return 2; }
	const uint64_t key = strtoull(argv[1], NULL, 0), gen = strtoull(argv[2], NULL, 0);
	const size_t dim = (size_t) atol(argv[3]), m = (size_t) atol(argv[4]), efc = (size_t) atol(argv[5]);
	const size_t efs = (size_t) atol(argv[6]);
	const int func = atoi(argv[7]);
	const char *qpath = argv[8];
	const size_t nq = (size_t) atol(argv[9]);
	const int nproc = atoi(argv[10]);
	const char *opath = argv[11];
	const int rounds = atoi(argv[12]);

	int qfd = open(qpath, O_RDONLY);
	if (qfd < 0) { perror(qpath); return 1; }
	const float *Q = (const float *) mmap(NULL, nq * dim * 4, PROT_READ, MAP_SHARED, qfd, 0);
	if (Q == MAP_FAILED) { perror("mmap queries"); return 1; }
	const size_t obytes = nq * efs * 8 + nq * 8;
	int ofd = open(opath, O_RDWR | O_CREAT | O_TRUNC, 0600);
	if (ofd < 0 || ftruncate(ofd, (off_t) obytes) != 0) { perror(opath); return 1; }
	uint64_t *out = (uint64_t *) mmap(NULL, obytes, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
	if (out == MAP_FAILED) { perror("mmap out"); return 1; }
	memset(out, 0xFF, nq * efs * 8);
	uint64_t *counts = out + nq * efs;

	int ready[2], go[2];
	if (pipe(ready) != 0 || pipe(go) != 0) { perror("pipe"); return 1; }
	for (int p = 0; p < nproc; p++)
	{
		pid_t pid = fork();
		if (pid < 0) { perror("fork"); return 1; }
		if (pid == 0)
		{
			close(ready[0]); close(go[1]);
			hnsw_init_dist_func();                                        /* _PG_init, embedding.c:150 */
			FlatIndex *f = flat_create(dim, m, efc, efs, func, 1);        /* meta only: the rows live in HBM */
			HnswMetadata *meta = flat_meta(f);
			uint64_t have = 0; int present = 0;
			if (hnsw_gpu_remote_lookup(key, &have, NULL, &present) != 0 || !present || have != gen ||
				hnsw_gpu_remote_attach(meta, key, gen) != 0)              /* hnsw_beginscan */
			{
				fprintf(stderr, "client %d: mirror %llx gen %llu not on the server: %s\n", p, (unsigned long long) key,
						(unsigned long long) gen, hnsw_gpu_remote_last_error());
				_exit(3);
			}
			char c = 'r';
			if (write(ready[1], &c, 1) != 1) _exit(4);
			if (read(go[0], &c, 1) != 1) _exit(4);
			const size_t q0 = nq * (size_t) p / (size_t) nproc, q1 = nq * (size_t) (p + 1) / (size_t) nproc;
			for (int r = 0; r < rounds; r++)
				for (size_t q = q0; q < q1; q++)
				{
					size_t n = 0;
					label_t *res = NULL;
					if (!hnsw_search(meta, Q + q * dim, &n, &res))        /* embedding.c:317 */
					{
						fprintf(stderr, "client %d: HNSW index search failed\n", p);
						_exit(5);
					}
					memcpy(out + q * efs, res, n * 8);
					counts[q] = n;
					free(res);                                            /* embedding.c:327 */
				}
			hnsw_gpu_remote_detach(meta);                                 /* hnsw_endscan */
			_exit(0);
		}
	}
	close(ready[1]); close(go[0]);
	for (int p = 0; p < nproc; p++)
	{
		char c;
		if (read(ready[0], &c, 1) != 1) { fprintf(stderr, "a client failed to start\n"); return 1; }
	}
	const double t0 = now_s();
	for (int p = 0; p < nproc; p++)
	{
		char c = 'g';
		if (write(go[1], &c, 1) != 1) { perror("go"); return 1; }
	}
	int bad = 0;
	for (int p = 0; p < nproc; p++)
	{
		int st = 0;
		if (wait(&st) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++;
	}
	const double sec = now_s() - t0;
	if (bad) { fprintf(stderr, "%d clients failed\n", bad); return 1; }
	msync(out, obytes, MS_SYNC);
	printf("{\"nproc\": %d, \"nq\": %zu, \"rounds\": %d, \"seconds\": %.6f, \"qps\": %.1f}\n", nproc, nq, rounds, sec,
		   (double) nq * rounds / sec);
	return 0;
}
