// Practical roof for the search kernel's access pattern: independent waves gathering RANDOM rows of a
// table (16-B loads, whole rows, T loads per lane in flight), no dependency between iterations.
// The search kernel cannot beat this at the same row size and occupancy; it can only approach it.
//   usage: gather_roof <rows> <dim> <loads_in_flight_per_lane> <waves_per_cu> [iters]
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/gather_roof scripts/micro/gather_roof.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

template <int T>
__global__ __launch_bounds__(256) void gather(const float4 *base, uint32_t nrows, uint32_t row_f4, uint32_t iters, float *out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	float acc = 0.f;
	uint32_t rr[T], oo[T];                              // row-in-iteration and float4-in-row of each of my loads
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
			const uint32_t row = __umulhi(mix(seed + rr[t]), nrows);
			v[t] = base[(size_t) row * row_f4 + oo[t]];
		}
#pragma unroll
		for (int t = 0; t < T; t++) acc += v[t].x + v[t].y + v[t].z + v[t].w;
	}
	if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv)
{
	const size_t rows = argc > 1 ? atoll(argv[1]) : 1000000;
	const uint32_t dim = argc > 2 ? atoi(argv[2]) : 768;
	const int T = argc > 3 ? atoi(argv[3]) : 24;
	const int wpc = argc > 4 ? atoi(argv[4]) : 8;
	const uint32_t iters = argc > 5 ? atoi(argv[5]) : 400;
	const uint32_t row_f4 = dim / 4;
	float4 *base; float *out;
	CK(hipMalloc(&base, rows * dim * 4)); CK(hipMalloc(&out, 4));
	CK(hipMemset(base, 0, rows * dim * 4));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const uint32_t blocks = p.multiProcessorCount * wpc / 4;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++)
	{
		CK(hipEventRecord(e0));
		switch (T)
		{
			case 4:  hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, base, (uint32_t) rows, row_f4, iters, out); break;
			case 8:  hipLaunchKernelGGL(gather<8>, dim3(blocks), dim3(256), 0, 0, base, (uint32_t) rows, row_f4, iters, out); break;
			case 12: hipLaunchKernelGGL(gather<12>, dim3(blocks), dim3(256), 0, 0, base, (uint32_t) rows, row_f4, iters, out); break;
			case 16: hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(256), 0, 0, base, (uint32_t) rows, row_f4, iters, out); break;
			case 24: hipLaunchKernelGGL(gather<24>, dim3(blocks), dim3(256), 0, 0, base, (uint32_t) rows, row_f4, iters, out); break;
			default: printf("T must be 4, 8, 12, 16 or 24\n"); return 1;
		}
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	const double bytes = (double) blocks * 4 * iters * T * 64 * 16;
	printf("rows %zu dim %u (row %u B, table %.2f GB) loads/lane %d waves/CU %d: %.3f ms  %.0f GB/s\n",
		   rows, dim, dim * 4, rows * dim * 4 / 1e9, T, wpc, best, bytes / best / 1e6);
	return 0;
}
