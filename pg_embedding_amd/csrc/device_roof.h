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
// T = 16-byte loads per lane in flight per pass (search kernel: KB * RPG: 24 at 768 dims).  A row is row_f4 float4 = lpr =
// ceil(row_f4 / 16) loads per lane of its 16-lane group.  T >= lpr: the 4 groups of a wave take 4 * (T / lpr) whole rows per
// pass; T < lpr: 4 rows per pass in lpr / T steps of T loads.  The host only picks T with T % lpr == 0 or lpr % T == 0, so
// every load of a pass is a load the trace asks for (no padding traffic, no partial rows).
// CHECK (tests only): also sums the bit patterns of every word the trace asks for (mod 2^64, order-free) into *check, so a test
// can tell that the replay read exactly the traced rows, whole.
template <int T, bool CHECK>
__global__ __launch_bounds__(256) void replay_roof_kernel(const float4 *__restrict__ base, uint32_t row_f4, const uint32_t *__restrict__ evals,
														   uint32_t evals_cap, const uint32_t *__restrict__ nevals, uint32_t nq,
														   uint32_t *ticket, float *out, unsigned long long *check)
{
	unsigned long long bits = 0;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t g = lane >> 4, sub = lane & 15;
	const uint32_t lpr = (row_f4 + 15) / 16;                 // loads per lane per row
	const uint32_t kb = lpr < (uint32_t) T ? lpr : (uint32_t) T;   // chunk-steps of one row per pass
	const uint32_t rpg = (uint32_t) T / kb;                  // rows per group per pass
	float acc = 0.f;
	for (;;)
	{
		uint32_t qi = 0;
		if (lane == 0) qi = atomicAdd(ticket, 1u);
		qi = __builtin_amdgcn_readfirstlane(qi);
		if (qi >= nq) break;
		const uint32_t *ids = evals + (size_t) qi * evals_cap;
		uint32_t ne = nevals[2 * (size_t) qi];               // (the search kernel's stats array: {evals, hops} per query)
		ne = ne < evals_cap ? ne : evals_cap;
		for (uint32_t r0 = 0; r0 < ne; r0 += 4 * rpg)
			for (uint32_t k0 = 0; k0 < lpr; k0 += kb)
			{
				float4 v[T];
#pragma unroll
				for (int t = 0; t < T; t++)
				{
					const uint32_t rr = (uint32_t) t / kb, c = (k0 + (uint32_t) t % kb) * 16 + sub;
					uint32_t r = r0 + rr * 4 + g;
					r = r < ne ? r : ne - 1;                 // (the last pass of a query re-reads its last row where the search kernel narrows the pass)
					const uint32_t row = ids[r];
					v[t] = base[(size_t) row * row_f4 + (c < row_f4 ? c : row_f4 - 1)];
					if (CHECK && r0 + rr * 4 + g < ne && c < row_f4)
						bits += (unsigned long long) __float_as_uint(v[t].x) + __float_as_uint(v[t].y) + __float_as_uint(v[t].z) + __float_as_uint(v[t].w);
				}
#pragma unroll
				for (int t = 0; t < T; t++) acc += (v[t].x + v[t].y) + (v[t].z + v[t].w);
			}
	}
	if (acc == 12345.678f) out[0] = acc;    // keeps the loads alive
	if (CHECK) atomicAdd(check, bits);
}

}  // namespace pgemb
