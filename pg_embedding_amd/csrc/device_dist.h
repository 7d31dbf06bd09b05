// device_dist.h — gfx950 distance primitives for the HNSW hot path.
//
// Replaces the per-pair loops of distfunc.c (l2 :28-65/:67-118/:121-130, cosine
// :133-145, manhattan :147-155) with a wave64 formulation:
//
//   * a wavefront is split into four 16-lane groups; each group scores ONE row, so
//     one `global_load_dwordx4` wave-instruction moves 4 rows x 256 contiguous bytes
//     (1 KiB, fully coalesced per row segment);
//   * lane `sub` of a group owns the float4 chunks sub, sub+16, sub+32, ... of the
//     row and of the LDS-staged query -> element e accumulates into partial sum
//     number e % 64, with ONE fused multiply-add per element;
//   * the 4 components are folded as (x+y)+(z+w) and the 16 lanes by an xor
//     butterfly 1,2,4,8 done with DPP row operations (no LDS traffic);
//   * epilogues exactly as the reference writes them (sqrtf / double-precision
//     1 - dot/sqrt(na*nb) / none).
//
// This fixes the summation ORDER.  oracle/hnsw_port.c restates the same order on the
// CPU, so oracle and device agree bit-for-bit; versus the reference's -Ofast build
// (whose order is compiler-chosen) the difference is round-off only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgemb {

enum : int { F_L2 = 0, F_COSINE = 1, F_MANHATTAN = 2 };   // embedding.h:22-26
// Opt-in arithmetic (HNSW_GPU_REF_ORDER=1, score_rows_ref below; all three functions): the summation order of oracle/_ref's OWN build of
// distfunc.c (gcc -Ofast, read off its disassembly), so that the device returns the compiled reference's id lists query by query —
// not just "equal except at near-ties" through the canonical-order oracle.  Since round 6 it loads rows exactly as the canonical code
// does and is a timed mode of bench.py; it stays opt-in because that order belongs to ONE compiler's output.
enum : int { F_L2_REF = 3, F_MANHATTAN_REF = 4, F_COSINE_REF = 5 };

// Compiler-level ordering point for cross-lane LDS hand-offs inside ONE wavefront
// (LDS operations of a wave execute in order; this only stops the compiler from
// moving memory operations across the hand-off).
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Sum over the 16 lanes of a DPP row; every lane of the row ends with the total.
// Order: xor 1, xor 2 (quad_perm), then row_half_mirror and row_mirror, which after
// the quads / octets are uniform deliver exactly the xor-4 / xor-8 partner's value.
__device__ __forceinline__ float row16_sum(float t)
{
	t = t + dpp_move<0xB1>(t);    // quad_perm [1,0,3,2]  == lane ^ 1
	t = t + dpp_move<0x4E>(t);    // quad_perm [2,3,0,1]  == lane ^ 2
	t = t + dpp_move<0x141>(t);   // row_half_mirror      == value of lane ^ 4
	t = t + dpp_move<0x140>(t);   // row_mirror           == value of lane ^ 8
	return t;
}

__device__ __forceinline__ float fold4(const float4 &a)
{
	return (a.x + a.y) + (a.z + a.w);
}

// Per-row running state: L2 / Manhattan use `a`; cosine uses a = dot, b = |x|^2.
struct RowAcc
{
	float4 a, b;
};

__device__ __forceinline__ void acc_zero(RowAcc &s)
{
	s.a = make_float4(0.f, 0.f, 0.f, 0.f);
	s.b = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int FUNC>
__device__ __forceinline__ void acc_step(RowAcc &s, const float4 &q, const float4 &x)
{
	if (FUNC == F_L2)
	{
		float d0 = q.x - x.x, d1 = q.y - x.y, d2 = q.z - x.z, d3 = q.w - x.w;
		s.a.x = __builtin_fmaf(d0, d0, s.a.x);
		s.a.y = __builtin_fmaf(d1, d1, s.a.y);
		s.a.z = __builtin_fmaf(d2, d2, s.a.z);
		s.a.w = __builtin_fmaf(d3, d3, s.a.w);
	}
	else if (FUNC == F_COSINE)
	{
		s.a.x = __builtin_fmaf(q.x, x.x, s.a.x);
		s.a.y = __builtin_fmaf(q.y, x.y, s.a.y);
		s.a.z = __builtin_fmaf(q.z, x.z, s.a.z);
		s.a.w = __builtin_fmaf(q.w, x.w, s.a.w);
		s.b.x = __builtin_fmaf(x.x, x.x, s.b.x);
		s.b.y = __builtin_fmaf(x.y, x.y, s.b.y);
		s.b.z = __builtin_fmaf(x.z, x.z, s.b.z);
		s.b.w = __builtin_fmaf(x.w, x.w, s.b.w);
	}
	else
	{
		s.a.x = s.a.x + __builtin_fabsf(q.x - x.x);
		s.a.y = s.a.y + __builtin_fabsf(q.y - x.y);
		s.a.z = s.a.z + __builtin_fabsf(q.z - x.z);
		s.a.w = s.a.w + __builtin_fabsf(q.w - x.w);
	}
}


// Epilogue of one distance from its reduced sums, exactly as the reference writes it
// (distfunc.c:64,117,129 / :144 / :154).  s0 = sum of squares | dot | sum of |.|; s1 = |x|^2
// (cosine only); qnorm = |q|^2 in the same canonical order.  Pure per-lane arithmetic, so callers
// run it once per row with one row per lane instead of once per 16-lane group: the correctly
// rounded sqrtf costs ~15 instructions and the fp64 divide/sqrt of the cosine form several
// times that.
template <int FUNC>
__device__ __forceinline__ float finish_dist(float s0, float s1, float qnorm)
{
	if (FUNC == F_L2 || FUNC == F_L2_REF)
		return __builtin_sqrtf(s0);
	if (FUNC == F_COSINE || FUNC == F_COSINE_REF)
	{
		const float prod = qnorm * s1;                                  // float product, distfunc.c:144
		const double r = 1.0 - (double) s0 / __builtin_sqrt((double) prod);
		return (float) r;
	}
	return s0;
}

// |q|^2 of the LDS-staged query, canonical order; all lanes return the same value.
__device__ __forceinline__ float query_norm(const float4 *q4, uint32_t nchunks, uint32_t kiters, int lane)
{
	const uint32_t sub = lane & 15;
	float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
	for (uint32_t k = 0; k < kiters; k++)
	{
		uint32_t c = k * 16 + sub;
		float4 q = (c < nchunks) ? q4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
		a.x = __builtin_fmaf(q.x, q.x, a.x);
		a.y = __builtin_fmaf(q.y, q.y, a.y);
		a.z = __builtin_fmaf(q.z, q.z, a.z);
		a.w = __builtin_fmaf(q.w, q.w, a.w);
	}
	return row16_sum(fold4(a));
}

// Score `nrows` rows against the LDS-staged query.
//   row r lives at vec + rowid(r) * stride (floats; stride % 4 == 0, zero padded)
//   q4     : query in LDS as float4 chunks, zero padded to a multiple of KB*16 chunks
//   out[r] : REDUCED SUM of row r (sum of squares | dot | sum of |.|), written by the first lane
//            of the owning group; cosine also writes |x|^2 to out[O2 + r] (O2 = OUT2 unless the caller's
//            array is a short one: the slice areas of the team form).  The caller turns the
//            sums into distances with finish_dist(), one row per lane.
// Shape <KB, RPG>: every 16-lane group owns RPG rows per pass (4*RPG rows per wave-pass) and
// issues KB chunk-steps of ALL its rows before the first use, i.e. KB*RPG independent
// 16-byte loads per lane (KB*RPG KiB per wave) are in flight per memory round trip.  The
// traversal is a chain of dependent round trips, so the shape is picked per dimensionality
// to cover a whole row per trip when it fits: 768 dims = <12,1>, 128 dims = <2,4>.
constexpr uint32_t OUT2 = 64;      // offset of the second sum (cosine |x|^2) in a score_rows output array

// The reference build's own order (HNSW_GPU_REF_ORDER=1: F_L2_REF / F_MANHATTAN_REF / F_COSINE_REF), as oracle/_ref/distfunc.o computes it:
//   l2_dist_impl_avx2 (distfunc.c:28-65, dims % 16 == 0): eight accumulators, acc_j += (d0_j^2 + d1_j^2) per 16 elements with
//     d0 = x[16k + j] - y[16k + j], d1 = x[16k + 8 + j] - y[16k + 8 + j] (two multiplies, one add, one add: no FMA), then
//     ((t0 + t1) + (t2 + t3)) + ((t6 + t7) + (t4 + t5)), sqrtf;
//   manhattan_dist_impl (distfunc.c:147-155, auto-vectorised, dims % 4 == 0): four accumulators acc_j += |x[4k + j] - y[4k + j]|,
//     then (a0 + a2) + (a1 + a3);
//   cosine_dist_impl (distfunc.c:133-145, auto-vectorised, dims % 4 == 0): three sets of four accumulators, dot_j += x[4k + j] * y[4k + j]
//     (mulps, addps: no FMA) and the two squared norms likewise, each reduced as (a0 + a2) + (a1 + a3); float product of the
//     norms, double 1 - dot / sqrt(product) (finish_dist).
// Round 6: the production LOAD shape with a TRANSPOSED accumulation (rounds 3-5: one lane per accumulator with strided scalar loads —
// "exists to be compared, not timed").  Each of those orders is a serial chain per accumulator over the WHOLE row, and a coalesced
// 16-byte load hands every lane four consecutive elements — four different accumulators.  So the rows are fetched exactly as the
// canonical code fetches them (lane `sub` of a 16-lane group owns float4 chunks sub, sub + 16, ...; KB chunk-steps of 2 rows per group
// in flight = 8 rows per memory round trip), the per-element terms (d^2 | q*x and x*x | |q - x|) are formed in that layout, and one
// 64-float slice of every row at a time goes through a per-wave LDS stage laid out BY ACCUMULATOR: lane (row slot = lane >> 3,
// j = lane & 7) then owns one accumulator of one row — L2: acc_j; cosine: j < 4 the dot chain's a_j, j >= 4 the norm chain's; Manhattan:
// j < 4 — reads its terms of the slice in the reference's order with 16-byte LDS reads and extends its chain with plain adds.  Memory
// traffic and bytes in flight equal the canonical path's; the price is 2-3x its VALU count per row plus ~1.5 KB of LDS traffic per row
// and slice, which a launch that is bound by HBM hides.  Padding chunks contribute exactly +0 (the query image is zero padded and the
// row chunk is zeroed), and x + 0 == x bit for bit for every value a chain can hold (a chain that starts at +0 never holds -0).
//   stage = the wave's LDS right behind its query image: 8 row slots of REF_STAGE_ROW (L2, Manhattan) or 2 x that (cosine) floats.
constexpr uint32_t REF_STAGE_ROW = 72;          // 64 terms + 8 floats of padding (row slots of the 4 groups start in different banks)

__device__ __forceinline__ float query_norm_ref(const float *qf, uint32_t n, int lane)      // |q|^2 in cosine_dist_impl's order; every lane returns it
{
	const uint32_t j = lane & 3;
	float acc = 0.f;
	for (uint32_t k = 0; k + 4 <= n; k += 4) { const float m = qf[k + j] * qf[k + j]; acc = acc + m; }
	acc = acc + dpp_move<0x4E>(acc);
	acc = acc + dpp_move<0xB1>(acc);
	return acc;
}

template <int FUNC, int KB, uint32_t O2, typename RowId>
__device__ __forceinline__ void score_rows_ref(const float *__restrict__ vec, size_t stride, const float4 *q4, uint32_t nchunks, uint32_t kiters,
											   RowId rowid, uint32_t nrows, float *out, int lane)
{
	constexpr int RPG = 2;                                       // 8 rows per pass = the 8 row slots of the stage
	constexpr uint32_t ROWF = FUNC == F_COSINE_REF ? 2 * REF_STAGE_ROW : REF_STAGE_ROW;
	const uint32_t g = lane >> 4, sub = lane & 15;
	const uint32_t slot_r = (uint32_t) lane >> 3, j = lane & 7; // this lane's accumulator in the transposed phase
	float *stage = const_cast<float *>(reinterpret_cast<const float *>(q4)) + (size_t) ((kiters + KB - 1) / KB * KB) * 64;
	const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
	for (uint32_t base = 0; base < nrows; base += 4 * RPG)
	{
		const float4 *row4[RPG];
#pragma unroll
		for (int rr = 0; rr < RPG; rr++)
		{
			const uint32_t r = base + rr * 4 + g;
			row4[rr] = reinterpret_cast<const float4 *>(vec + (size_t) rowid(r < nrows ? r : nrows - 1) * stride);
		}
		float acc = 0.f;
		for (uint32_t k0 = 0; k0 < kiters; k0 += KB)
		{
			float4 x[RPG][KB];
			if ((k0 + KB) * 16 <= nchunks)              // wave-uniform: whole batch inside the row
			{
#pragma unroll
				for (int u = 0; u < KB; u++)
#pragma unroll
					for (int rr = 0; rr < RPG; rr++) x[rr][u] = row4[rr][(k0 + u) * 16 + sub];
			}
			else
			{
#pragma unroll
				for (int u = 0; u < KB; u++)
				{
					const uint32_t c = (k0 + u) * 16 + sub;
					const uint32_t cc = c < nchunks ? c : nchunks - 1;
#pragma unroll
					for (int rr = 0; rr < RPG; rr++)
					{
						const float4 t = row4[rr][cc];
						x[rr][u] = c < nchunks ? t : zero4;
					}
				}
			}
#pragma unroll
			for (int u = 0; u < KB; u++)
			{
				if ((k0 + u) * 16 >= nchunks) break;            // (wave-uniform) slices past the row's end hold nothing but zeros
				const float4 q = q4[(k0 + u) * 16 + sub];       // LDS image is zero padded
				// ---- terms of this 64-float slice, in the load layout, to their accumulators' places in the stage ----
#pragma unroll
				for (int rr = 0; rr < RPG; rr++)
				{
					float *srow = stage + (size_t) (rr * 4 + g) * ROWF;
					const float4 xv = x[rr][u];
					if (FUNC == F_L2_REF)
					{
						// element e = 4 * sub + comp of the slice: block kk = sub >> 2, half h = (sub >> 1) & 1, accumulator j = 4 * (sub & 1) + comp;
						// accumulator j's eight terms [kk][h] are contiguous
						const float d0 = q.x - xv.x, d1 = q.y - xv.y, d2 = q.z - xv.z, d3 = q.w - xv.w;
						float *p = srow + (4u * (sub & 1u)) * 8u + (sub >> 2) * 2u + ((sub >> 1) & 1u);
						p[0] = d0 * d0; p[8] = d1 * d1; p[16] = d2 * d2; p[24] = d3 * d3;
					}
					else if (FUNC == F_COSINE_REF)
					{
						// accumulator `comp` of the dot chain / of the norm chain: sixteen terms (chunks sub = 0..15) contiguous each
						float *p = srow + sub;
						p[0] = q.x * xv.x; p[16] = q.y * xv.y; p[32] = q.z * xv.z; p[48] = q.w * xv.w;
						p[64] = xv.x * xv.x; p[80] = xv.y * xv.y; p[96] = xv.z * xv.z; p[112] = xv.w * xv.w;
					}
					else
					{
						float *p = srow + sub;
						p[0] = __builtin_fabsf(q.x - xv.x); p[16] = __builtin_fabsf(q.y - xv.y);
						p[32] = __builtin_fabsf(q.z - xv.z); p[48] = __builtin_fabsf(q.w - xv.w);
					}
				}
				wave_sync();
				// ---- every accumulator extends its chain by its terms of the slice, in the reference's order ----
				{
					const float *mine = stage + (size_t) slot_r * ROWF;
					if (FUNC == F_L2_REF)
					{
						const float4 a = *reinterpret_cast<const float4 *>(mine + j * 8u), b = *reinterpret_cast<const float4 *>(mine + j * 8u + 4u);
						acc = acc + (a.x + a.y); acc = acc + (a.z + a.w); acc = acc + (b.x + b.y); acc = acc + (b.z + b.w);
					}
					else
					{
						// cosine: j < 4 = dot accumulator j, j >= 4 = norm accumulator j - 4 (both sets of 64 floats); Manhattan: lanes j >= 4 idle along
						const float *t = mine + (FUNC == F_COSINE_REF ? j : (j & 3u)) * 16u;
#pragma unroll
						for (int i = 0; i < 4; i++)
						{
							const float4 v = *reinterpret_cast<const float4 *>(t + 4 * i);
							acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w;
						}
					}
				}
				wave_sync();                                    // the stage is rewritten by the next slice
			}
		}
		// ---- the reference's horizontal sums (all lanes: DPP reads neighbours) ----
		const uint32_t r = base + slot_r;
		if (FUNC == F_L2_REF)
		{
			acc = acc + dpp_move<0xB1>(acc);      // t0+t1 | t2+t3 | t4+t5 | t6+t7
			acc = acc + dpp_move<0x4E>(acc);      // (t0+t1)+(t2+t3) | (t4+t5)+(t6+t7) (= (t6+t7)+(t4+t5) bit for bit)
			acc = acc + dpp_move<0x141>(acc);     // the two halves of the eight lanes
			if (j == 0 && r < nrows) out[r] = acc;
		}
		else
		{
			acc = acc + dpp_move<0x4E>(acc);      // a0+a2 | a1+a3
			acc = acc + dpp_move<0xB1>(acc);      // (a0+a2)+(a1+a3)
			if (j == 0 && r < nrows) out[r] = acc;
			if (FUNC == F_COSINE_REF && j == 4 && r < nrows) out[O2 + r] = acc;
		}
	}
}

template <int FUNC, int KB, int RPG, uint32_t O2 = OUT2, typename RowId>
__device__ __forceinline__ void score_rows(const float *__restrict__ vec, size_t stride,
										   const float4 *q4, uint32_t nchunks, uint32_t kiters,
										   RowId rowid, uint32_t nrows, float *out, int lane)
{
	if (FUNC == F_L2_REF || FUNC == F_MANHATTAN_REF || FUNC == F_COSINE_REF)
	{
		score_rows_ref<FUNC, KB, O2>(vec, stride, q4, nchunks, kiters, rowid, nrows, out, lane);
		return;
	}
	const uint32_t g = lane >> 4, sub = lane & 15;
	const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
	// Every load below is UNCONDITIONAL: rows past the end re-read the last valid row and
	// chunks past the end re-read the last valid chunk, and the unwanted values are replaced
	// by a select afterwards.  (A load under a per-lane `if` makes hipcc branch around it and
	// wait vmcnt(0) right behind it, which serialises the whole batch.)
	for (uint32_t base = 0; base < nrows; base += 4 * RPG)
	{
		const float4 *row4[RPG];
		bool v[RPG];
		RowAcc s[RPG];
#pragma unroll
		for (int rr = 0; rr < RPG; rr++)
		{
			const uint32_t r = base + rr * 4 + g;
			v[rr] = r < nrows;
			row4[rr] = reinterpret_cast<const float4 *>(vec + (size_t) rowid(v[rr] ? r : nrows - 1) * stride);
			acc_zero(s[rr]);
		}
		for (uint32_t k0 = 0; k0 < kiters; k0 += KB)
		{
			float4 x[RPG][KB];
			if ((k0 + KB) * 16 <= nchunks)          // wave-uniform: whole batch inside the row
			{
#pragma unroll
				for (int u = 0; u < KB; u++)
#pragma unroll
					for (int rr = 0; rr < RPG; rr++) x[rr][u] = row4[rr][(k0 + u) * 16 + sub];
			}
			else
			{
#pragma unroll
				for (int u = 0; u < KB; u++)
				{
					const uint32_t c = (k0 + u) * 16 + sub;
					const uint32_t cc = c < nchunks ? c : nchunks - 1;
#pragma unroll
					for (int rr = 0; rr < RPG; rr++)
					{
						const float4 t = row4[rr][cc];
						x[rr][u] = c < nchunks ? t : zero4;
					}
				}
			}
			// Query chunks come from LDS in groups of QG while the row loads are in flight;
			// the schedule is pinned so that hipcc does not hoist all KB LDS reads above the
			// arithmetic (that costs 4*KB more live VGPRs and a wave of occupancy).
			constexpr int QG = KB < 4 ? KB : 4;
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u0 = 0; u0 < KB; u0 += QG)
			{
				float4 q[QG];
#pragma unroll
				for (int j = 0; j < QG; j++)
					if (u0 + j < KB) q[j] = q4[(k0 + u0 + j) * 16 + sub];      // LDS image is zero padded
#pragma unroll
				for (int j = 0; j < QG; j++)
					if (u0 + j < KB)
					{
#pragma unroll
						for (int rr = 0; rr < RPG; rr++) acc_step<FUNC>(s[rr], q[j], x[rr][u0 + j]);
					}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
#pragma unroll
		for (int rr = 0; rr < RPG; rr++)
		{
			const float s0 = row16_sum(fold4(s[rr].a));             // all 64 lanes: DPP reads neighbours
			float s1 = 0.f;
			if (FUNC == F_COSINE) s1 = row16_sum(fold4(s[rr].b));
			const uint32_t r = base + rr * 4 + g;
			if (sub == 0 && v[rr])
			{
				out[r] = s0;
				if (FUNC == F_COSINE) out[O2 + r] = s1;
			}
		}
	}
}

// score_rows with the LAST pass narrowed to the rows that are left: full passes of 4*RPG rows, then one
// pass with RPG/2 or RPG/4 rows per group when that covers the remainder.  A hop yields 5-9 new rows on
// average, so without this most passes issue (clamped, L1-hit) loads and arithmetic for absent rows.
// The per-row summation order does not depend on RPG, so results are unchanged bit for bit.
template <int FUNC, int KB, int RPG, uint32_t O2 = OUT2, typename RowId>
__device__ __forceinline__ void score_rows_fit(const float *__restrict__ vec, size_t stride,
											   const float4 *q4, uint32_t nchunks, uint32_t kiters,
											   RowId rowid, uint32_t nrows, float *out, int lane)
{
	if (FUNC == F_L2_REF || FUNC == F_MANHATTAN_REF || FUNC == F_COSINE_REF)      // (its own passes of 8 rows, whatever the shape's RPG)
	{
		score_rows<FUNC, KB, RPG, O2>(vec, stride, q4, nchunks, kiters, rowid, nrows, out, lane);
		return;
	}
	const uint32_t full = nrows / (4 * RPG) * (4 * RPG);
	if (full) score_rows<FUNC, KB, RPG, O2>(vec, stride, q4, nchunks, kiters, rowid, full, out, lane);
	const uint32_t rem = nrows - full;
	if (rem == 0) return;
	auto shifted = [rowid, full](uint32_t r) { return rowid(full + r); };
	if (RPG >= 4 && rem > 8)
		score_rows<FUNC, KB, RPG, O2>(vec, stride, q4, nchunks, kiters, shifted, rem, out + full, lane);
	else if (RPG >= 2 && rem > 4)
		score_rows<FUNC, KB, 2, O2>(vec, stride, q4, nchunks, kiters, shifted, rem, out + full, lane);
	else
		score_rows<FUNC, KB, 1, O2>(vec, stride, q4, nchunks, kiters, shifted, rem, out + full, lane);
}

// Load-batch shapes by chunk-steps per row (kiters = ceil(dim/64)).
struct Shape2x4  { static constexpr int KB = 2,  RPG = 4, MIN_WAVES = 4; };   // dim <= 128
struct Shape2x2  { static constexpr int KB = 2,  RPG = 2, MIN_WAVES = 5; };   // dim <= 128, the hot beam form: 96 VGPRs, 5 waves/SIMD
struct Shape4x2  { static constexpr int KB = 4,  RPG = 2, MIN_WAVES = 4; };   // dim <= 256
struct Shape8x2  { static constexpr int KB = 8,  RPG = 2, MIN_WAVES = 2; };   // dim <= 512
// (experiment builds: -DHNSW_W3_12X1 / -DHNSW_W3_12X2 cap that shape's kernels at 168 VGPRs = 3 waves/SIMD; the host side
// then wants HNSW_GPU_WIDE_WAVES=12 and HNSW_GPU_TEAM_WPB=6 or 4)
#ifdef HNSW_W3_12X1
struct Shape12x1 { static constexpr int KB = 12, RPG = 1, MIN_WAVES = 3; };
#else
struct Shape12x1 { static constexpr int KB = 12, RPG = 1, MIN_WAVES = 2; };   // larger (768 = one batch)
#endif
#ifdef HNSW_W3_12X2
struct Shape12x2 { static constexpr int KB = 12, RPG = 2, MIN_WAVES = 3; };
#else
struct Shape12x2 { static constexpr int KB = 12, RPG = 2, MIN_WAVES = 2; };   // same, 8 rows per round trip
#endif

__host__ __device__ inline int shape_index(uint32_t kiters)
{
	return kiters <= 2 ? 0 : kiters <= 4 ? 1 : kiters <= 8 ? 2 : 3;
}
__host__ __device__ inline uint32_t shape_kb(int idx)
{
	return idx == 0 ? 2u : idx == 1 ? 4u : idx == 2 ? 8u : 12u;
}

}  // namespace pgemb
