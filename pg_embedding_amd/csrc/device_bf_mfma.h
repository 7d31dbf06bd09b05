// device_bf_mfma.h — exhaustive scoring as a dense contraction on the matrix cores.
//
// Graph search scores a private candidate list per query (a batch of GEMVs: HBM-bound, no MFMA
// claim).  EXHAUSTIVE scoring of a query batch against every row is different: Q x D times D x N
// is a genuine GEMM (BASELINE config 5, recall ground truth), bound by the f32 MFMA roof
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s on MI355X), not by HBM.
//
// Exactness is kept by using the GEMM only as a FILTER:
//   1. a per-query bound tau_q = k-th smallest CANONICAL distance (device_dist.h) over a
//      sample of rows — an upper bound of the true k-th distance;
//   2. this kernel computes all Q x N dot products with MFMA (a k-ordered fused-multiply-add
//      chain: exact f32, but a different summation order than the canonical one), turns them
//      into approximate distances and appends every row within tau_q (+ a round-off margin)
//      to the query's candidate list;
//   3. the few survivors are re-scored with the canonical code and the top-k is taken from
//      those distances — identical to the all-canonical brute force, ties by lower idx.
// L2 uses |q|^2 + |x|^2 - 2 q.x, cosine 1 - q.x / sqrt(|q|^2 |x|^2) (distfunc.c:133-145);
// Manhattan is not a contraction and is not offered here.
//
// Tiling (round 5): block = 4 waves, 128 queries x 128 rows per block, K in steps of 32 floats.  Global loads are whole
// 128-byte lines (8 lanes x float4 per tile row: one wave instruction = 8 rows x 128 B) and go to LDS ROW-major with a padded
// row stride of 36 floats, one ds_write_b128 per float4 — no transposition: a filter may sum a dot product in any k order as
// long as both operands use the same one, so lane (col, kk) of an MFMA takes the four k of ONE ds_read_b128 (k = 8g + 4kk + s,
// s = 0..3) and feeds them to four consecutive MFMAs.  Per 32-float K step a wave issues 16 ds_read_b128 for 64 MFMAs (round
// 1-4: 64 ds_read_b32 + 32 ds_write_b32 per thread, and global loads that touched 64 different lines per instruction).  Two
// LDS buffers: the stores of step ks + 1 go to the other buffer, ONE barrier per step.  Each wave owns a 64 x 64 sub-tile =
// 2 x 2 MFMA tiles (64 accumulator registers).  blockIdx is remapped so that all query tiles of one row tile run on the same
// XCD (block b -> XCD b % 8) and the row tile is fetched from HBM once per XCD L2.
#pragma once
#include "device_search.h"

namespace pgemb {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));       // (a first-class vector: HIP's float4 struct went through scratch here)

constexpr int BF_TQ = 128, BF_TR = 128, BF_TK = 32;
#ifndef BF_GLDS
#define BF_GLDS 1                                    // 1: tiles go global -> LDS directly (global_load_lds_dwordx4), 0: through registers
#endif
#ifndef BF_NBUF
#define BF_NBUF 2
#endif
#if BF_GLDS
constexpr int BF_LS = BF_TK;                         // lane-linear LDS image: 128-byte rows, chunk c of row r at slot c ^ ((r >> 1) & 7)
#else
constexpr int BF_LS = BF_TK + 4;                     // padded rows: 144 B, 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots
#endif
constexpr int BF_TILE_FLOATS = BF_TQ * BF_LS;        // one operand tile
constexpr int BF_EPI_FLOATS = 2 * BF_TQ;              // the tile's per-query bound and |q|^2, staged for the epilogue
constexpr size_t BF_LDS_BYTES = ((size_t) BF_NBUF * 2 * BF_TILE_FLOATS + BF_EPI_FLOATS) * sizeof(float);

struct BfArgs
{
	const float *queries;      // [nq][qstride] copy, zero padded to whole K steps (qstride = round_up(stride, BF_TK))
	const float *qnorm;        // |q|^2
	const float *qbound;       // L2: squared bound; cosine: threshold on dot / sqrt(|x|^2)
	const float *vec;          // [n][stride]
	const float *xnorm;        // |x|^2
	uint32_t nq, n, stride, qstride, ksteps;
	int func;
	uint32_t *cand;            // [nq][cap]
	uint32_t *cand_cnt;        // [nq]
	uint32_t cap;
	uint32_t nqt, nrt;         // tiles
	unsigned long long *clocks; // NULL, or 2 words: shader-clock and constant-clock ticks one block spent in its K loop (measurement only)
};

__global__ __launch_bounds__(256, 2) void bf_mfma_filter_kernel(const BfArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float *bf_lds = reinterpret_cast<float *>(smem);                    // [buf][A | B][row][BF_LS], then the epilogue's bounds
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	// XCD-aware tile order: the nqt query tiles of one row tile share b % 8
	const uint32_t b = blockIdx.x;
	const uint32_t xcd = b & 7, rest = b >> 3;
	const uint32_t qt = rest % a.nqt, rgrp = rest / a.nqt;
	const uint32_t rt = rgrp * 8 + xcd;
	if (rt >= a.nrt) return;
	const uint32_t q0 = qt * BF_TQ, r0 = rt * BF_TR;

	// staging role: thread owns one 16-byte slot `sch` (of the 8 of a K step) of tile rows srow + 32 j: a wave instruction reads
	// 8 rows x 128 contiguous bytes
	const uint32_t sch = t & 7, srow = t >> 3;
	const uint32_t nchunks = a.stride / 4;
	const floatx4 *qsrc[4], *xsrc[4];
#pragma unroll
	for (int j = 0; j < 4; j++)
	{
		qsrc[j] = reinterpret_cast<const floatx4 *>(a.queries + (size_t) min(q0 + srow + 32 * j, a.nq - 1) * a.qstride);
		xsrc[j] = reinterpret_cast<const floatx4 *>(a.vec + (size_t) min(r0 + srow + 32 * j, a.n - 1) * a.stride);
	}

	floatx16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
#pragma unroll
			for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
	const uint32_t wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves over the block tile
	const uint32_t kk = lane >> 5, col = lane & 31;

	// What the epilogue compares with is fetched NOW, behind the K loop: every block of a launch takes the same time, so the blocks of
	// a CU reach their epilogues together, and an epilogue that loads its 32 bounds per lane one dependent L2 round trip after the
	// other (rounds 1-4) leaves the matrix pipe idle for a tenth of every round.  The tile's 128 bounds and |q|^2 go to LDS (read back
	// four at a time), the two |x|^2 of the lane's columns to registers.
	float *epi = bf_lds + (size_t) BF_NBUF * 2 * BF_TILE_FLOATS;
	if (t < BF_TQ)
	{
		const uint32_t qi = min(q0 + (uint32_t) t, a.nq - 1);
		epi[t] = a.qbound[qi];
		epi[BF_TQ + t] = a.qnorm[qi];
	}
	float xs2[2];
#pragma unroll
	for (int j = 0; j < 2; j++)
	{
		const float xn = a.xnorm[min(r0 + wn * 64 + j * 32 + col, a.n - 1)];
		xs2[j] = (a.func == F_COSINE) ? __builtin_sqrtf(xn) : xn;
	}

	// No select on a loaded value (it would pull the wait for the loads in front of the MFMAs): the query copy is zero padded
	// to whole K steps, and a row chunk beyond the row's end re-reads the row's last chunk (times zero: nothing).
#if BF_GLDS
	// The LDS image is what the hardware writes: wave-uniform base + lane * 16, i.e. 8 rows x 128 B per instruction, no padding.
	// Bank conflicts are avoided on the SOURCE side: slot p of row r holds chunk p ^ ((r >> 1) & 7) — still whole 128-byte lines per
	// row from memory, and the 16 lanes of every ds_read_b128 group ((r & 1), (r >> 1) & 7 all distinct) land on 16 distinct slots.
	const uint32_t gch = sch ^ ((srow >> 1) & 7);              // the chunk this lane fetches ((srow + 32 j) >> 1) & 7 == (srow >> 1) & 7
	auto fetch = [&](uint32_t ks, uint32_t buf)
	{
		const uint32_t c = ks * 8 + gch;
		const uint32_t cc = min(c, nchunks - 1);
		float *As = bf_lds + (size_t) buf * 2 * BF_TILE_FLOATS + (wave * 8) * BF_LS, *Bs = As + BF_TILE_FLOATS;
#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (qsrc[j] + c),
											 (__attribute__((address_space(3))) void *) (As + 32 * j * BF_LS), 16, 0, 0);
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (xsrc[j] + cc),
											 (__attribute__((address_space(3))) void *) (Bs + 32 * j * BF_LS), 16, 0, 0);
		}
	};
	const uint32_t swz = (col >> 1) & 7;
	uint32_t roff[BF_TK / 8];                                  // float offset of k-group g's slot in this lane's row
#pragma unroll
	for (int g = 0; g < BF_TK / 8; g++) roff[g] = ((2 * g + kk) ^ swz) * 4;
#else
	floatx4 qa[4], xb[4];
	auto fetch = [&](uint32_t ks)
	{
		const uint32_t c = ks * 8 + sch;                          // float4 chunk along K
		const uint32_t cc = min(c, nchunks - 1);
#pragma unroll
		for (int j = 0; j < 4; j++) { qa[j] = qsrc[j][c]; xb[j] = xsrc[j][cc]; }
	};
	auto stage = [&](uint32_t buf)
	{
		float *As = bf_lds + (size_t) buf * 2 * BF_TILE_FLOATS, *Bs = As + BF_TILE_FLOATS;
#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			*reinterpret_cast<floatx4 *>(As + (srow + 32 * j) * BF_LS + sch * 4) = qa[j];
			*reinterpret_cast<floatx4 *>(Bs + (srow + 32 * j) * BF_LS + sch * 4) = xb[j];
		}
	};
#endif
	auto contract = [&](uint32_t buf)
	{
		const float *As = bf_lds + (size_t) buf * 2 * BF_TILE_FLOATS + (wm * 64 + col) * BF_LS;
		const float *Bs = bf_lds + (size_t) buf * 2 * BF_TILE_FLOATS + BF_TILE_FLOATS + (wn * 64 + col) * BF_LS;
#pragma unroll
		for (int g = 0; g < BF_TK / 8; g++)
		{
#if BF_GLDS
			const uint32_t o = roff[g];
#else
			const uint32_t o = g * 8 + kk * 4;
#endif
			const floatx4 a0 = *reinterpret_cast<const floatx4 *>(As + o);
			const floatx4 a1 = *reinterpret_cast<const floatx4 *>(As + 32 * BF_LS + o);
			const floatx4 b0 = *reinterpret_cast<const floatx4 *>(Bs + o);
			const floatx4 b1 = *reinterpret_cast<const floatx4 *>(Bs + 32 * BF_LS + o);
#define BF_STEP(C)                                                                          \
			acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b0.C, acc[0][0], 0, 0, 0);    \
			acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b1.C, acc[0][1], 0, 0, 0);    \
			acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b0.C, acc[1][0], 0, 0, 0);    \
			acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b1.C, acc[1][1], 0, 0, 0);
			BF_STEP(x) BF_STEP(y) BF_STEP(z) BF_STEP(w)
#undef BF_STEP
		}
	};

	unsigned long long c0 = 0, r0c = 0;
	if (a.clocks) { c0 = __builtin_readcyclecounter(); r0c = wall_clock64(); }
	// Software pipeline: the loads of K step ks + 1 are issued before the 64 MFMAs of step ks.
#if BF_GLDS && BF_NBUF == 2
	fetch(0, 0);
	__syncthreads();                                            // (hipcc drains the LDS-bound loads, vmcnt(0), in front of the barrier)
	for (uint32_t ks = 0; ks < a.ksteps; ks++)
	{
		fetch(min(ks + 1, a.ksteps - 1), (ks + 1) & 1);            // the other buffer: its readers passed the previous barrier
		__builtin_amdgcn_sched_barrier(0);
		contract(ks & 1);
		__syncthreads();
	}
#elif BF_GLDS
	for (uint32_t ks = 0; ks < a.ksteps; ks++)
	{
		__syncthreads();                                            // previous step's operand reads are done
		fetch(ks, 0);
		__syncthreads();
		contract(0);
	}
#elif BF_NBUF == 2
	fetch(0);
	stage(0);
	__syncthreads();
	for (uint32_t ks = 0; ks < a.ksteps; ks++)
	{
		fetch(min(ks + 1, a.ksteps - 1));                          // (branch-free: the last step re-reads itself into the idle buffer)
		__builtin_amdgcn_sched_barrier(0);                         // the loads are ISSUED here, not sunk behind the MFMAs
		contract(ks & 1);
		__builtin_amdgcn_sched_barrier(0);
		stage((ks + 1) & 1);                                       // the other buffer: its readers passed the previous barrier
		__syncthreads();
	}
#else
	fetch(0);
	for (uint32_t ks = 0; ks < a.ksteps; ks++)
	{
		__syncthreads();                                            // previous step's operand reads are done
		stage(0);
		__syncthreads();
		if (ks + 1 < a.ksteps) fetch(ks + 1);                       // in flight during the MFMAs below
		contract(0);
	}
#endif
	if (a.clocks && blockIdx.x == gridDim.x / 2 && t == 0)             // a block from the middle of the launch
	{
		a.clocks[0] = __builtin_readcyclecounter() - c0;
		a.clocks[1] = wall_clock64() - r0c;
	}

	// epilogue: C[q][r]; lane holds column r = lane & 31, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
	// (the K loop's barriers lie between the stores of `epi` and these reads)
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int e4 = 0; e4 < 4; e4++)
		{
			const uint32_t ql = wm * 64 + i * 32 + 8 * e4 + 4 * kk;          // four consecutive queries of the tile
			const floatx4 qb = *reinterpret_cast<const floatx4 *>(epi + ql);
			const floatx4 qn = *reinterpret_cast<const floatx4 *>(epi + BF_TQ + ql);
#pragma unroll
			for (int j = 0; j < 2; j++)
			{
				const uint32_t r = r0 + wn * 64 + j * 32 + col;
				const bool rok = r < a.n;
#pragma unroll
				for (int e1 = 0; e1 < 4; e1++)
				{
					const uint32_t q = q0 + ql + e1;
					const float dot = acc[i][j][e4 * 4 + e1];
					bool pass;
					if (a.func == F_COSINE)
						pass = dot >= qb[e1] * xs2[j];                   // 1 - dot/sqrt(nq nx) <= tau (+margin)
					else
						pass = qn[e1] + xs2[j] - 2.f * dot <= qb[e1];    // |q-x|^2 <= tau^2 (+margin)
					if (pass && rok && q < a.nq)
					{
						const uint32_t pos = atomicAdd(&a.cand_cnt[q], 1u);
						if (pos < a.cap) a.cand[(size_t) q * a.cap + pos] = r;
					}
				}
			}
		}
}

// |row|^2 for every row (plain accumulation; only used by the filter and its margin)
__global__ void row_norm2_kernel(const float *__restrict__ vec, uint32_t n, uint32_t stride, float *__restrict__ out)
{
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	if (w >= n) return;
	const float4 *p = reinterpret_cast<const float4 *>(vec + (size_t) w * stride);
	float s = 0.f;
	for (uint32_t c = lane; c < stride / 4; c += 64)
	{
		const float4 v = p[c];
		s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
	}
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
	if (lane == 0) out[w] = s;
}

// queries [nq][dim] -> zero padded [nq][stride]
__global__ void pad_queries_kernel(const float *__restrict__ q, uint32_t nq, uint32_t dim, uint32_t stride, float *__restrict__ out)
{
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t) nq * stride) return;
	const uint32_t r = (uint32_t) (i / stride), c = (uint32_t) (i % stride);
	out[i] = c < dim ? q[(size_t) r * dim + c] : 0.f;
}

// tau_q (canonical k-th distance over the sample) -> the filter's comparison value with margin
__global__ void make_bounds_kernel(const float *__restrict__ tau, const float *__restrict__ qnorm, uint32_t nq, int func,
								   float *__restrict__ qbound)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nq) return;
	const float t = tau[q];
	if (func == F_COSINE)
	{
		// pass if dot >= (1 - tau - eps) * sqrt(nq) * sqrt(nx); eps covers both summation orders
		qbound[q] = (1.f - t - 2e-5f - 1e-5f * __builtin_fabsf(t)) * __builtin_sqrtf(qnorm[q]);
		if (!(t == t)) qbound[q] = -__builtin_inff();               // NaN bound: keep everything
	}
	else
	{
		// pass if |q|^2 + |x|^2 - 2 dot <= tau^2 (1 + eps) + eps' (|q|^2 + ...): generous
		qbound[q] = t * t * (1.f + 1e-4f) + 1e-5f * (qnorm[q] + t * t) + 1e-12f;
	}
}

// One wave per query: canonical distances of the surviving rows, top-k by (dist, idx).
template <int FUNC>
__global__ __launch_bounds__(256) void bf_rescore_kernel(const float *__restrict__ vec, uint32_t dim, uint32_t stride,
														 uint32_t nchunks, uint32_t kiters, uint32_t qpad_floats,
														 const float *__restrict__ queries, uint32_t nq,
														 const uint32_t *__restrict__ cand, const uint32_t *__restrict__ cand_cnt,
														 uint32_t cap, uint32_t k, uint32_t *__restrict__ out_idx,
														 float *__restrict__ out_dist, uint32_t *__restrict__ overflow)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const uint32_t qi = blockIdx.x * 4 + wib;
	if (qi >= nq) return;
	const size_t wave_bytes = (size_t) qpad_floats * 4 + (size_t) (k + 1) * 8 + 128 * 4;
	unsigned char *my = smem + wib * ((wave_bytes + 15) & ~(size_t) 15);
	float *qf = reinterpret_cast<float *>(my);
	const float4 *q4 = reinterpret_cast<const float4 *>(my);
	uint64_t *top = reinterpret_cast<uint64_t *>(my + (size_t) qpad_floats * 4);
	float *sums = reinterpret_cast<float *>(top + (k + 1));
	for (uint32_t e = lane; e < qpad_floats; e += 64)
	{
		const float t = queries[(size_t) qi * dim + (e < dim ? e : dim - 1)];
		qf[e] = e < dim ? t : 0.f;
	}
	wave_sync();
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, nchunks, kiters, lane);
	uint32_t cnt = cand_cnt[qi];
	if (cnt > cap) { if (lane == 0) atomicAdd(overflow, 1u); cnt = cap; }
	const uint32_t *ids = cand + (size_t) qi * cap;
	uint32_t tsize = 0;
	uint64_t worst = ~0ull;
	for (uint32_t base = 0; base < cnt; base += 64)
	{
		const uint32_t c64 = min(64u, cnt - base);
		auto by_id = [ids, base](uint32_t r) { return ids[base + r]; };
		score_rows<FUNC, 4, 2>(vec, stride, q4, nchunks, kiters, by_id, c64, sums, lane);
		wave_sync();
		const float dl = finish_dist<FUNC>(sums[lane], sums[OUT2 + lane], qnorm);
		const uint64_t kl = ((uint64_t) ord_f32(dl) << 32) | ids[base + ((uint32_t) lane < c64 ? lane : 0)];
		uint64_t todo = __ballot((uint32_t) lane < c64 && (tsize < k || kl < worst));
		while (todo)
		{
			const uint32_t r = (uint32_t) __builtin_ctzll(todo);
			todo &= todo - 1;
			const uint64_t key = readlane_u64(kl, r);
			if (tsize < k || key < worst)
			{
				tsize = sorted_insert(top, tsize, key, k, lane);
				worst = top[tsize - 1];
			}
		}
		wave_sync();
	}
	for (uint32_t i = lane; i < k; i += 64)
	{
		const bool ok = i < tsize;
		out_idx[(size_t) qi * k + i] = ok ? (uint32_t) top[i] : LINK_NONE;
		if (out_dist) out_dist[(size_t) qi * k + i] = ok ? unord_f32((uint32_t) (top[i] >> 32)) : __builtin_inff();
	}
}

}  // namespace pgemb
