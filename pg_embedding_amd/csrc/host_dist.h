// host_dist.h — ONE distance on the host, in the canonical summation order of the gfx950 kernels.
//
// hnsw_dist_func (distfunc.c:171-174) has two callers in the reference: the search loop
// (hnswalg.cpp:39 — replaced by the fused device kernels, which never call this symbol) and the SQL
// operators <-> / <=> / <~>, which hand over ONE pair per fmgr call (calc_distance,
// embedding.c:1022-1046).  One pair is ~3 KB of arithmetic: a kernel launch for it costs 16-28 us
// against ~0.1 us on the calling core, so this symbol is computed where the caller already is.
// It is NOT a fallback for the device path (search, insert and the batch entry points have none and
// fail without a device); it is the whole implementation of the one-pair symbol.
//
// Bit-identical to device_dist.h (tests/test_gpu_dropin.py, tests/test_abi.py):
//   * element e accumulates into partial sum e % 64 with ONE fused multiply-add (v_fma_f32 there,
//     vfmadd here; without FMA hardware: fmaf(), which rounds once as well);
//   * t[l] = (s[4l] + s[4l+1]) + (s[4l+2] + s[4l+3]), l = 0..15                  (fold4)
//   * xor butterfly over the 16 t's in the order 1, 2, 4, 8                       (row16_sum)
//   * epilogues as the reference writes them: sqrtf (distfunc.c:64,117,129); float product of the two
//     norms, then double-precision 1 - dot / sqrt(prod) (:144); none (:154).
// Build with -ffp-contract=off: nothing here may be fused except the explicit FMAs.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <immintrin.h>

namespace hostdist {

enum : int { H_L2 = 0, H_COSINE = 1, H_MANHATTAN = 2 };     // dist_func_t, embedding.h:22-26

static inline float reduce64(const float *s)
{
	float t[16], u[16];
	for (int l = 0; l < 16; l++) t[l] = (s[4 * l] + s[4 * l + 1]) + (s[4 * l + 2] + s[4 * l + 3]);
	for (int off = 1; off < 16; off <<= 1)
	{
		for (int l = 0; l < 16; l++) u[l] = t[l] + t[l ^ off];
		memcpy(t, u, sizeof(t));
	}
	return t[0];
}

static inline float epilogue(int func, float s0, float s1, float qn)
{
	if (func == H_L2) return sqrtf(s0);
	if (func == H_COSINE)
	{
		const float prod = qn * s1;                                   // float product, distfunc.c:144
		const double r = 1.0 - (double) s0 / sqrt((double) prod);
		return (float) r;
	}
	return s0;
}

// Portable form (any x86-64 / any host): fmaf() is correctly rounded with or without FMA hardware.
static float dist_scalar(int func, const float *q, const float *x, size_t dim)
{
	float s0[64], s1[64], s2[64];
	for (int j = 0; j < 64; j++) s0[j] = s1[j] = s2[j] = 0.f;
	for (size_t e = 0; e < dim; e++)
	{
		const int j = (int) (e & 63);
		const float a = q[e], b = x[e];
		if (func == H_L2) { const float d = a - b; s0[j] = fmaf(d, d, s0[j]); }
		else if (func == H_COSINE) { s0[j] = fmaf(a, b, s0[j]); s1[j] = fmaf(b, b, s1[j]); s2[j] = fmaf(a, a, s2[j]); }
		else s0[j] = s0[j] + fabsf(a - b);
	}
	if (func == H_COSINE) return epilogue(func, reduce64(s0), reduce64(s1), reduce64(s2));
	return epilogue(func, reduce64(s0), 0.f, 0.f);
}

#if defined(__x86_64__)
// fold4 + the xor butterfly on eight registers r[v] = s[8v .. 8v+7].  hadd(hadd(a,b), hadd(c,d)) is exactly
// (x0+x1)+(x2+x3) per 4-group, landing as U0 = [t0 t2 t4 t6 | t1 t3 t5 t7], U1 = the same for t8..t15; the
// butterfly partners are then: xor 1 = the other 128-bit half, xor 2 = the neighbour, xor 4 = two over,
// xor 8 = the other register.  Every stage adds all 16 lanes at once, like row16_sum on the device.
__attribute__((target("avx2,fma"))) static inline float reduce64_avx2(const __m256 *r)
{
	__m256 u0 = _mm256_hadd_ps(_mm256_hadd_ps(r[0], r[1]), _mm256_hadd_ps(r[2], r[3]));
	__m256 u1 = _mm256_hadd_ps(_mm256_hadd_ps(r[4], r[5]), _mm256_hadd_ps(r[6], r[7]));
	u0 = _mm256_add_ps(u0, _mm256_permute2f128_ps(u0, u0, 1));
	u1 = _mm256_add_ps(u1, _mm256_permute2f128_ps(u1, u1, 1));
	u0 = _mm256_add_ps(u0, _mm256_permute_ps(u0, 0xB1));
	u1 = _mm256_add_ps(u1, _mm256_permute_ps(u1, 0xB1));
	u0 = _mm256_add_ps(u0, _mm256_permute_ps(u0, 0x4E));
	u1 = _mm256_add_ps(u1, _mm256_permute_ps(u1, 0x4E));
	return _mm_cvtss_f32(_mm256_castps256_ps128(_mm256_add_ps(u0, u1)));
}

// the last, partial 64-element step: through memory, scalar FMAs into the partial sums it touches
template <int WHICH>        // 0: (a-b)^2   1: a*b   2: b*b   3: a*a   4: |a-b|
__attribute__((target("avx2,fma"))) static inline void tail_avx2(__m256 *r, const float *q, const float *x, size_t n)
{
	alignas(32) float s[64];
	for (int v = 0; v < 8; v++) _mm256_store_ps(s + 8 * v, r[v]);
	for (size_t j = 0; j < n; j++)
	{
		const float a = q[j], b = x[j];
		if (WHICH == 0) { const float d = a - b; s[j] = __builtin_fmaf(d, d, s[j]); }
		else if (WHICH == 1) s[j] = __builtin_fmaf(a, b, s[j]);
		else if (WHICH == 2) s[j] = __builtin_fmaf(b, b, s[j]);
		else if (WHICH == 3) s[j] = __builtin_fmaf(a, a, s[j]);
		else s[j] = s[j] + __builtin_fabsf(a - b);
	}
	for (int v = 0; v < 8; v++) r[v] = _mm256_load_ps(s + 8 * v);
}

// AVX2 + FMA form: the 64 partial sums are eight 8-float registers; 64 elements per step.
__attribute__((target("avx2,fma"))) static float l2_avx2(const float *q, const float *x, size_t dim)
{
	__m256 r[8];
	for (int v = 0; v < 8; v++) r[v] = _mm256_setzero_ps();
	size_t e = 0;
	for (; e + 64 <= dim; e += 64)
		for (int v = 0; v < 8; v++)
		{
			const __m256 d = _mm256_sub_ps(_mm256_loadu_ps(q + e + 8 * v), _mm256_loadu_ps(x + e + 8 * v));
			r[v] = _mm256_fmadd_ps(d, d, r[v]);
		}
	if (e < dim) tail_avx2<0>(r, q + e, x + e, dim - e);
	return sqrtf(reduce64_avx2(r));
}

__attribute__((target("avx2,fma"))) static float manhattan_avx2(const float *q, const float *x, size_t dim)
{
	__m256 r[8];
	for (int v = 0; v < 8; v++) r[v] = _mm256_setzero_ps();
	const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF));
	size_t e = 0;
	for (; e + 64 <= dim; e += 64)
		for (int v = 0; v < 8; v++)
		{
			const __m256 d = _mm256_sub_ps(_mm256_loadu_ps(q + e + 8 * v), _mm256_loadu_ps(x + e + 8 * v));
			r[v] = _mm256_add_ps(r[v], _mm256_and_ps(d, absmask));
		}
	if (e < dim) tail_avx2<4>(r, q + e, x + e, dim - e);
	return reduce64_avx2(r);
}

// cosine: three sums; 24 accumulators exceed the 16 ymm registers, so the row is walked once per pair of
// sums (dot + |x|^2, then |q|^2) — the summation order of each sum is what matters, not the interleaving.
__attribute__((target("avx2,fma"))) static float cosine_avx2(const float *q, const float *x, size_t dim)
{
	__m256 rd[8], rb[4], rb2[4];
	for (int v = 0; v < 8; v++) rd[v] = _mm256_setzero_ps();
	for (int v = 0; v < 4; v++) rb[v] = rb2[v] = _mm256_setzero_ps();
	size_t e = 0;
	const size_t full = dim / 64 * 64;
	for (; e < full; e += 64)
	{
		for (int v = 0; v < 4; v++)
		{
			const __m256 a = _mm256_loadu_ps(q + e + 8 * v), b = _mm256_loadu_ps(x + e + 8 * v);
			rd[v] = _mm256_fmadd_ps(a, b, rd[v]);
			rb[v] = _mm256_fmadd_ps(b, b, rb[v]);
		}
		for (int v = 4; v < 8; v++)
		{
			const __m256 a = _mm256_loadu_ps(q + e + 8 * v), b = _mm256_loadu_ps(x + e + 8 * v);
			rd[v] = _mm256_fmadd_ps(a, b, rd[v]);
			rb2[v - 4] = _mm256_fmadd_ps(b, b, rb2[v - 4]);
		}
	}
	__m256 rx[8], ra[8];
	for (int v = 0; v < 4; v++) { rx[v] = rb[v]; rx[v + 4] = rb2[v]; }
	for (int v = 0; v < 8; v++) ra[v] = _mm256_setzero_ps();
	for (e = 0; e < full; e += 64)
		for (int v = 0; v < 8; v++)
		{
			const __m256 a = _mm256_loadu_ps(q + e + 8 * v);
			ra[v] = _mm256_fmadd_ps(a, a, ra[v]);
		}
	if (full < dim)
	{
		tail_avx2<1>(rd, q + full, x + full, dim - full);
		tail_avx2<2>(rx, q + full, x + full, dim - full);
		tail_avx2<3>(ra, q + full, x + full, dim - full);
	}
	return epilogue(H_COSINE, reduce64_avx2(rd), reduce64_avx2(rx), reduce64_avx2(ra));
}
#endif

// dispatch of distfunc.c:171-174 (unknown values score as Manhattan there: table of three; here NaN)
static inline float dist(int func, const float *q, const float *x, size_t dim)
{
	if (func < 0 || func > 2) return NAN;
#if defined(__x86_64__)
	static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
	if (fast) return func == H_L2 ? l2_avx2(q, x, dim) : func == H_COSINE ? cosine_avx2(q, x, dim) : manhattan_avx2(q, x, dim);
#endif
	return dist_scalar(func, q, x, dim);
}

}  // namespace hostdist
