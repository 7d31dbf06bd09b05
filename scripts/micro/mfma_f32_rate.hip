// What does v_mfma_f32_32x32x2_f32 sustain on THIS device, from registers alone?  (round 5: the filter kernel of csrc/device_bf_mfma.h
// stays at 0.86-0.87 of the nominal f32 matrix roof whatever its tiling, buffering, occupancy or operand schedule.)
// usage: mfma_f32_rate [waves_per_simd=1] [data: 0 zeros | 1 random]      build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHARED>
__global__ __launch_bounds__(256) void burn(const float *in, float *out, int iters, unsigned long long *clk)
{
	const int t = threadIdx.x + blockIdx.x * blockDim.x;
	const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	float a[8], b[8];
	for (int i = 0; i < 8; i++) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + i + 4096) & 65535]; }
	floatx16 acc[4];
	for (int k = 0; k < 4; k++) for (int e = 0; e < 16; e++) acc[k][e] = 0.f;
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int u = 0; u < 8; u++)
		{
			if (SHARED)     // the filter kernel's pattern: a 2 x 2 tile, operands shared between neighbouring MFMAs
			{
				acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
				acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
				acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[2], 0, 0, 0);
				acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[(u + 1) & 7], acc[3], 0, 0, 0);
			}
			else
			{
				acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
				acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[(u + 2) & 7], acc[1], 0, 0, 0);
				acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 3) & 7], b[(u + 4) & 7], acc[2], 0, 0, 0);
				acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 5) & 7], b[(u + 6) & 7], acc[3], 0, 0, 0);
			}
		}
	}
	float s = 0.f;
	for (int k = 0; k < 4; k++) for (int e = 0; e < 16; e++) s += acc[k][e];
	out[t] = s;
	if (t == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }
}

int main(int argc, char **argv)
{
	const int wps = argc > 1 ? atoi(argv[1]) : 1, rnd = argc > 2 ? atoi(argv[2]) : 1;
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, blocks = cus * wps, iters = 20000;
	std::vector<float> h(65536);
	for (size_t i = 0; i < h.size(); i++) h[i] = rnd ? (float) rand() / RAND_MAX * 2.f - 1.f : 0.f;
	float *in, *out; hipMalloc(&in, 65536 * 4); hipMalloc(&out, (size_t) blocks * 256 * 4);
	unsigned long long *clk, hclk[2]; hipMalloc(&clk, 16);
	int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
	hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int shared = 0; shared < 2; shared++)
		for (int rep = 0; rep < 3; rep++)
		{
			hipEventRecord(e0);
			if (shared) hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
			else hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
			const double flops = (double) blocks * 4 * iters * 32 * 4096.0;
			printf("%d CUs, %d wave(s)/SIMD, %s data, %s operands: %.2f ms, %.1f TFLOP/s = %.3f of 157.3, shader clock %.0f MHz\n", cus, wps, rnd ? "random" : "zero",
				   shared ? "2x2-shared" : "distinct", ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, khz * 1e-3 * (double) hclk[0] / (double) hclk[1]);
		}
	return 0;
}
