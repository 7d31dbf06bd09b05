// gpu_diag.hip — measurement only (include/hnsw_gpu_diag.h): traced launches, team counters, the roofs bench.py prices the search kernel against
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"
#include "device_roof.h"

// The same launch as hnsw_gpu_search_batch_dev that also writes its evaluation trace: d_evals[i * evals_cap + j] = the j-th row
// query i scored (j < d_stats[2 * i], truncated at evals_cap), d_times[2 * i], [2 * i + 1] = the device's constant-rate clock
// (100 MHz) at the start of query i and at the end of its walk.  Measurement only (bench.py: replay roof, reuse distances).
extern "C" int hnsw_gpu_search_traced_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										  label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										  idx_t *d_evals, size_t evals_cap, uint64_t *d_times, void *stream)
{
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (!d_evals || evals_cap == 0 || evals_cap > 0xFFFFFFFFull || !d_stats) return fail(HNSW_GPU_ERR_ARG, "trace buffers missing");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	ix->ws.evals_next = d_evals; ix->ws.evals_cap_next = (uint32_t) evals_cap; ix->ws.times_next = d_times;
	const int rc = launch_search(ix, &ix->ws, d_queries, ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats, (hipStream_t) stream);
	ix->ws.evals_next = nullptr; ix->ws.evals_cap_next = 0; ix->ws.times_next = nullptr;
	return rc;
}

extern "C" int hnsw_gpu_team_counters(hnsw_gpu_index *ix, uint32_t *out8)
{
	if (!ix || !out8) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	memset(out8, 0, 64);
	if (!ix->ws.team_dbg) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(out8, ix->ws.team_dbg, 64, hipMemcpyDeviceToHost));
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// measured roof of the access pattern (device_roof.h)
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_gather_roof(hnsw_gpu_index *ix, int loads_per_lane, int waves_per_cu, unsigned iters, float *gbps)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !gbps) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ix->n == 0 || iters == 0 || waves_per_cu <= 0 || (waves_per_cu & 3)) return fail(HNSW_GPU_ERR_ARG, "need rows, iters > 0 and waves_per_cu %% 4 == 0");
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t row_f4 = ix->stride / 4;
	const uint32_t blocks = (uint32_t) (ix->num_cu * waves_per_cu / 4);
	float *out = (float *) ix->misc + 8;
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	float best = 1e30f;
	int rc = HNSW_GPU_OK;
	for (int rep = 0; rep < 4 && rc == HNSW_GPU_OK; rep++)
	{
		(void) hipEventRecord(e0, nullptr);
		const float4 *base = (const float4 *) ix->vec;
		switch (loads_per_lane)
		{
#define ROOF(T) case T: hipLaunchKernelGGL(gather_roof_kernel<T>, dim3(blocks), dim3(256), 0, nullptr, base, (uint32_t) ix->n, row_f4, iters, out); break
			ROOF(4); ROOF(8); ROOF(12); ROOF(16); ROOF(24);
#undef ROOF
			default: rc = fail(HNSW_GPU_ERR_ARG, "loads_per_lane must be 4, 8, 12, 16 or 24");
		}
		if (rc) break;
		(void) hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(HNSW_GPU_ERR_HIP, "gather roof kernel failed"); break; }
		float ms = 0.f;
		(void) hipEventElapsedTime(&ms, e0, e1);
		if (rep > 0 && ms < best) best = ms;          // first repetition warms up
	}
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	if (rc) return rc;
	const double bytes = (double) blocks * 4.0 * iters * loads_per_lane * 64.0 * 16.0;
	*gbps = (float) (bytes / best / 1e6);
	return HNSW_GPU_OK;
}

// Replay roof (device_roof.h): the rows a traced launch scored, gathered again by `slots` resident waves in the same query
// order with nothing in between.  d_stats = that launch's stats array ({evals, hops} per query).  *ms = best of 3 timed
// repetitions (after one warm-up), *bytes = row bytes one repetition reads.
extern "C" int hnsw_gpu_replay_roof_parts(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
										  unsigned slots, int kb, int rpg, unsigned parts, float *ms, double *bytes, uint64_t *word_sum);
extern "C" int hnsw_gpu_replay_roof(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
									unsigned slots, int kb, int rpg, float *ms, double *bytes, uint64_t *word_sum)
{
	return hnsw_gpu_replay_roof_parts(ix, d_evals, evals_cap, d_stats, nq, slots, kb, rpg, 1, ms, bytes, word_sum);
}

// The same with every query's trace cut into `parts` equal pieces gathered by different waves: the roof of a launch that gives one
// walk's rows to `parts` waves (fewer queries than resident waves).
extern "C" int hnsw_gpu_replay_roof_parts(hnsw_gpu_index *ix, const idx_t *d_evals, size_t evals_cap, const uint32_t *d_stats, size_t nq,
										  unsigned slots, int kb, int rpg, unsigned parts, float *ms, double *bytes, uint64_t *word_sum)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_evals || !d_stats || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ix->n == 0 || nq == 0 || slots < 4 || evals_cap == 0) return fail(HNSW_GPU_ERR_ARG, "need rows, queries and at least 4 slots");
	if (parts == 0 || parts > 64 || nq * (size_t) parts >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "parts must be 1..64");
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t row_f4 = ix->stride / 4;
	const uint32_t blocks = slots / 4;
	float *out = (float *) ix->misc + 8;
	uint32_t *ticket = ix->misc + 12;
	unsigned long long *d_check = (unsigned long long *) (ix->misc + 14);
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	float best = 1e30f;
	int rc = HNSW_GPU_OK;
	const int shape = kb * 100 + rpg;
	for (int rep = 0; rep < 4 && rc == HNSW_GPU_OK; rep++)
	{
		(void) hipMemsetAsync(ticket, 0, 16, nullptr);           // ticket + the (test-only) word sum behind it
		(void) hipEventRecord(e0, nullptr);
		const float4 *base = (const float4 *) ix->vec;
		switch (shape)
		{
#define ROOF(K, R) case K * 100 + R: \
				if (word_sum) hipLaunchKernelGGL((replay_roof_kernel<K, R, true>), dim3(blocks), dim3(256), 4 * REPLAY_STAGE * 4, nullptr, base, row_f4, d_evals, (uint32_t) evals_cap, d_stats, (uint32_t) nq, (uint32_t) parts, ticket, out, d_check); \
				else hipLaunchKernelGGL((replay_roof_kernel<K, R, false>), dim3(blocks), dim3(256), 4 * REPLAY_STAGE * 4, nullptr, base, row_f4, d_evals, (uint32_t) evals_cap, d_stats, (uint32_t) nq, (uint32_t) parts, ticket, out, d_check); \
				break
			ROOF(2, 2); ROOF(2, 4); ROOF(2, 8); ROOF(4, 2); ROOF(4, 4); ROOF(8, 2); ROOF(12, 1); ROOF(12, 2); ROOF(6, 4);
#undef ROOF
			default: rc = fail(HNSW_GPU_ERR_ARG, "no replay shape <%d, %d> (have <2,2> <2,4> <2,8> <4,2> <4,4> <8,2> <6,4> <12,1> <12,2>)", kb, rpg);
		}
		if (rc) break;
		(void) hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(HNSW_GPU_ERR_HIP, "replay roof kernel failed"); break; }
		float t = 0.f;
		(void) hipEventElapsedTime(&t, e0, e1);
		if (rep > 0 && t < best) best = t;            // first repetition warms up
	}
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	if (rc) return rc;
	*ms = best;
	if (word_sum) HIPCHK(hipMemcpy(word_sum, d_check, 8, hipMemcpyDeviceToHost));   // of the last repetition
	if (bytes)
	{
		// rows actually in the trace: sum over queries of min(evals, cap)
		std::vector<uint32_t> st(2 * nq);
		HIPCHK(hipMemcpy(st.data(), d_stats, 2 * nq * 4, hipMemcpyDeviceToHost));
		double rows = 0;
		for (size_t i = 0; i < nq; i++) rows += (double) std::min<size_t>(st[2 * i], evals_cap);
		*bytes = rows * ix->stride * 4.0;
	}
	return HNSW_GPU_OK;
}


// ------------------------------------------------------------------------------------
// the clock a search launch ran at, and where the mirror sits (profiles/r5af_*: a launch state that depends on the process's history)
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_last_search_clock_mhz(hnsw_gpu_index *ix, double *mhz)
{
	if (!ix || !mhz) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	*mhz = 0.0;
	if (ix->ws.launches == 0) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipEventSynchronize(ix->ws.ev1[(ix->ws.launches - 1) % SearchWs::EV_RING]));
	uint64_t c[4] = { 0, 0, 0, 0 };
	HIPCHK(hipMemcpy(c, ix->ws.health + HEALTH_CLOCK, sizeof(c), hipMemcpyDeviceToHost));
	int khz = 0;
	if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ix->device) != hipSuccess || khz <= 0) khz = 100000;
	if (c[2] > c[0] && c[3] > c[1]) *mhz = khz * 1e-3 * (double) (c[2] - c[0]) / (double) (c[3] - c[1]);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_placement(hnsw_gpu_index *ix, uint64_t *out16)
{
	if (!ix || !out16) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	const SearchWs &w = ix->ws;
	const uint64_t v[16] = {
		(uint64_t) (uintptr_t) ix->arena, ix->arena_bytes,
		(uint64_t) (uintptr_t) ix->vec, ix->cap * ix->stride * sizeof(float),
		(uint64_t) (uintptr_t) ix->links, ix->cap * ix->lstride * sizeof(uint32_t),
		(uint64_t) (uintptr_t) ix->labels, ix->cap * sizeof(uint64_t),
		(uint64_t) (uintptr_t) w.vis, w.vis_slots * w.vis_words * 4,
		(uint64_t) (uintptr_t) w.vlog, w.vis_slots * (size_t) w.logcap * 4,
		(uint64_t) (uintptr_t) w.beam, w.beam_keys * 8,
		(uint64_t) (uintptr_t) w.ticket, 64 };
	memcpy(out16, v, sizeof(v));
	return HNSW_GPU_OK;
}
