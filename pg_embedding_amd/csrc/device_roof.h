// device_roof.h — the practical roof of the search kernel's access pattern, measured on the index's own
// row table: independent waves gathering RANDOM whole rows with 16-byte loads, T loads per lane in
// flight, nothing depending on anything.  The fused search kernel reads the same rows the same way but
// behind a dependent chain (pop -> links -> visited -> rows -> accept), so at equal row width it cannot
// beat this from HBM; whatever it achieves above it is served by the Infinity Cache / L2 (rows that
// several queries of one launch share).  bench.py reports it as roofline.measured_gather_GBps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgemb {

__device__ __forceinline__ uint32_t roof_mix(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

template <int T>
__global__ __launch_bounds__(256) void gather_roof_kernel(const float4 *__restrict__ base, uint32_t nrows, uint32_t row_f4,
														  uint32_t iters, float *out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	float acc = 0.f;
	uint32_t rr[T], oo[T];                  // row-of-this-iteration and float4-in-row of each of my loads
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		const uint32_t j = lane + 64u * t;
		rr[t] = j / row_f4;
		oo[t] = j % row_f4;
	}
	for (uint32_t it = 0; it < iters; it++)
	{
		float4 v[T];
		const uint32_t seed = wave * 0x9e3779b9u + it * 64u;
#pragma unroll
		for (int t = 0; t < T; t++)
		{
			const uint32_t row = __umulhi(roof_mix(seed + rr[t]), nrows);
			v[t] = base[(size_t) row * row_f4 + oo[t]];
		}
#pragma unroll
		for (int t = 0; t < T; t++) acc += (v[t].x + v[t].y) + (v[t].z + v[t].w);
	}
	if (acc == 12345.678f) out[0] = acc;    // keeps the loads alive
}

// Replay roof: the rows ONE search launch scored (its per-query evaluation trace, hnsw_gpu_search_traced_dev), gathered again
// by the same number of resident waves in the same query order (an atomic ticket, as the search kernel) with the search
// kernel's own load shape — whole rows, 16 lanes per row, 16-byte loads, as many rows per pass as cover the same bytes in
// flight per wave — and NOTHING in between: no pop, no link list, no visited test, no accept loop; the addresses of
// pass k+1 do not depend on the data of pass k.  Same bytes, same temporal locality between the queries of a launch
// (what the caches see is the same), no dependencies: search time / replay time is what the walk's dependent chain costs,
// replay bytes / replay time is what the memory system gives this trace.  The search kernel cannot beat it.
// Load shape <KB, RPG> as in score_rows (device_dist.h): every 16-lane group owns RPG rows per pass and issues KB chunk-steps
// of all of them before the first use = KB * RPG independent 16-byte loads per lane in flight (the search kernel's shape at
// 768 dims: <12, 2>); a row of lpr = ceil(row_f4 / 16) loads per lane takes ceil(lpr / KB) steps.  No address arithmetic beyond
// the search kernel's own: the replay must be limited by the memory system, not by integer divisions.
// CHECK (tests only): also sums the bit patterns of every word the trace asks for (mod 2^64, order-free) into *check, so a test
// can tell that the replay read exactly the traced rows, whole.
constexpr uint32_t REPLAY_STAGE = 512;
template <int KB, int RPG, bool CHECK>
__global__ __launch_bounds__(256) void replay_roof_kernel(const float4 *__restrict__ base, uint32_t row_f4, const uint32_t *__restrict__ evals,
														   uint32_t evals_cap, const uint32_t *__restrict__ nevals, uint32_t nq, uint32_t parts,
														   uint32_t *ticket, float *out, unsigned long long *check)
{
	// row ids staged in LDS, REPLAY_STAGE at a time (the search kernel has its ids in LDS too: a pass must not wait for an id
	// load before it can issue its row loads); dynamic LDS: 4 waves x REPLAY_STAGE x 4 bytes
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr uint32_t STAGE = REPLAY_STAGE;
	unsigned long long bits = 0;
	const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
	const uint32_t g = lane >> 4, sub = lane & 15;
	const uint32_t lpr = (row_f4 + 15) / 16;                 // loads per lane per row
	uint32_t *ids = reinterpret_cast<uint32_t *>(smem) + (size_t) wib * STAGE;
	float acc = 0.f;
	for (;;)
	{
		// (parts > 1: a query's trace is cut into `parts` equal pieces that different waves gather — the roof of a launch in which
		// `parts` waves share one walk's rows, i.e. of fewer queries than resident waves)
		uint32_t tk = 0;
		if (lane == 0) tk = atomicAdd(ticket, 1u);
		tk = __builtin_amdgcn_readfirstlane(tk);
		if (tk >= nq * parts) break;
		const uint32_t qi = tk / parts, part = tk - qi * parts;
		uint32_t ne = nevals[2 * (size_t) qi];               // (the search kernel's stats array: {evals, hops} per query)
		ne = ne < evals_cap ? ne : evals_cap;
		const uint32_t lo = (uint32_t) ((uint64_t) ne * part / parts), hi = (uint32_t) ((uint64_t) ne * (part + 1) / parts);
		const uint32_t *gids = evals + (size_t) qi * evals_cap + lo;
		ne = hi - lo;
		for (uint32_t s0 = 0; s0 < ne; s0 += STAGE)
		{
			const uint32_t ns = ne - s0 < STAGE ? ne - s0 : STAGE;
			__builtin_amdgcn_wave_barrier();
			for (uint32_t i = lane; i < ns; i += 64) ids[i] = gids[s0 + i];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			for (uint32_t r0 = 0; r0 < ns; r0 += 4 * RPG)
			{
				const float4 *rp[RPG];
				bool rin[RPG];
#pragma unroll
				for (int rr = 0; rr < RPG; rr++)
				{
					const uint32_t r = r0 + (uint32_t) rr * 4 + g;
					rin[rr] = r < ns;                        // (a partial last pass re-reads its last row where the search kernel narrows the pass)
					rp[rr] = base + (size_t) ids[rin[rr] ? r : ns - 1] * row_f4;
				}
				for (uint32_t k0 = 0; k0 < lpr; k0 += KB)
				{
					float4 x[RPG][KB];
#pragma unroll
					for (int u = 0; u < KB; u++)
					{
						const uint32_t c = (k0 + (uint32_t) u) * 16 + sub;
						const uint32_t cc = c < row_f4 ? c : row_f4 - 1;
#pragma unroll
						for (int rr = 0; rr < RPG; rr++)
						{
							x[rr][u] = rp[rr][cc];
							if (CHECK && rin[rr] && c < row_f4)
								bits += (unsigned long long) __float_as_uint(x[rr][u].x) + __float_as_uint(x[rr][u].y) + __float_as_uint(x[rr][u].z) + __float_as_uint(x[rr][u].w);
						}
					}
#pragma unroll
					for (int u = 0; u < KB; u++)
#pragma unroll
						for (int rr = 0; rr < RPG; rr++) acc += (x[rr][u].x + x[rr][u].y) + (x[rr][u].z + x[rr][u].w);
				}
			}
		}
	}
	if (acc == 12345.678f) out[0] = acc;    // keeps the loads alive
	if (CHECK) atomicAdd(check, bits);
}

}  // namespace pgemb
