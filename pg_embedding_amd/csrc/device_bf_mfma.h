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
// Tiling (round 5): block = 4 waves, 128 queries x 128 rows per block (8 waves and 256 x 256 for launches with tiles enough: BfTile below), K in
// steps of BF_TK floats.  Tiles go from global memory
// straight into LDS (global_load_lds_dwordx4) in whole 128-byte lines, bank-swizzled on the source side; operands are read with
// ds_read_b128 (four k per read: a filter may sum in any k order), 16 reads per 64 MFMAs, conflict-free (SQ_LDS_BANK_CONFLICT 0); two
// LDS buffers, one barrier per step; each wave owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles (64 accumulator registers); the epilogue's
// operands are fetched behind the K loop.  blockIdx is remapped so that all query tiles of one row tile run on the same XCD
// (block b -> XCD b % 8) and the row tile is fetched from HBM once per XCD L2.  (Rounds 1-4: K-major transposing ds_write_b32 staging
// from loads that touched 64 lines per instruction, 64 ds_read_b32 per 64 MFMAs: 122 TFLOP/s; this form: 134-135, profiles/r5*.)
#pragma once
#include "device_search.h"

namespace pgemb {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));       // (a first-class vector: HIP's float4 struct went through scratch here)

#ifndef BF_TK
#define BF_TK 32                                     // floats of K per step (32 or 64)
#endif
#define BF_NBUF 2                                    // LDS buffers: the loads of step ks + 1 fly during the MFMAs of step ks
#ifndef BF_ABLATE
#define BF_ABLATE 0                                  // 0 = the product.  1: no barrier in the K loop; 2: one operand read per K step; 3: no tile loads after the first
#endif
#ifndef BF_NJ
#define BF_NJ 2                                      // 32-row MFMA tiles of index rows per wave (2: 64 x 64 per wave, 4: 64 x 128)
#endif
#ifndef BF_WM
#define BF_WM 2                                      // waves along the queries (2: 128-query tiles, 4 waves; 4: 256-query tiles, 8 waves)
#endif
#ifndef BF_BIG
#define BF_BIG 1                                     // 1: launches with enough tiles use 256 x 256 block tiles (8 waves, 64 x 128 per wave, one block per CU)
#endif
constexpr int BF_CH = BF_TK / 4;                     // 16-byte chunks of a tile row per K step (8 or 16)
constexpr int BF_RPI = 64 / BF_CH;                   // tile rows one wave instruction fills (8 x 128 B or 4 x 256 B: whole lines either way)
constexpr int BF_LS = BF_TK;                         // lane-linear LDS image, no padding: chunk c of row r at slot c ^ swizzle(r)
constexpr int BF_PASS_CAP = 508;                     // (query, row) pairs of a block that passed the filter, collected in LDS (+ 4 words of counter: 4 KB)
// One block tile: WM x 2 waves of 64 queries x (32 NJ) rows each.  Two are instantiated: <BF_WM, BF_NJ> = 128 x 128 (4 waves, two blocks per
// CU) and <4, 4> = 256 x 256 (8 waves, one block per CU: half the tile loads per flop — 138 against 136 TFLOP/s — for launches with enough tiles).
template <int WM, int NJ>
struct BfTile
{
	static constexpr int WAVES = 2 * WM, THREADS = 64 * WAVES;
	static constexpr int TQ = 64 * WM, TR = 64 * NJ;        // block tile
	static constexpr int RPP = WAVES * BF_RPI;              // tile rows the block fills per pass
	static constexpr int PASSES = TQ / RPP;                 // load instructions per query tile, thread and K step
	static constexpr int PASSES_R = TR / RPP;               // ... per row tile
	static constexpr int TILE_FLOATS = TQ * BF_LS;          // the query tile
	static constexpr int BUF_FLOATS = (TQ + TR) * BF_LS;    // one buffer: query tile, then row tile
	static constexpr int EPI_FLOATS = 2 * TQ;               // the tile's per-query bound and |q|^2, staged for the epilogue
	static constexpr size_t LDS_BYTES = ((size_t) BF_NBUF * BUF_FLOATS + EPI_FLOATS + 4 + 2 * BF_PASS_CAP) * sizeof(float);
};
// bank swizzle of a tile row: a ds_read_b128 serves 16 lanes per cycle over a 256-byte bank row.  128-byte rows (BF_TK 32) put two
// rows in a bank row: slot = c ^ ((r >> 1) & 7) makes (r & 1, (r >> 1) & 7) — all distinct within a lane group — pick 16 distinct
// slots; 256-byte rows (BF_TK 64) fill one: slot = c ^ (r & 15).
__device__ __forceinline__ uint32_t bf_swz(uint32_t r) { return BF_TK == 32 ? ((r >> 1) & 7u) : (r & 15u); }

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

template <int WM, int NJ>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void bf_mfma_filter_kernel(const BfArgs a)
{
	using T = BfTile<WM, NJ>;
	constexpr int BF_THREADS = T::THREADS, BF_TQ = T::TQ, BF_TR = T::TR, BF_RPP = T::RPP, BF_PASSES = T::PASSES, BF_PASSES_R = T::PASSES_R;
	constexpr int BF_TILE_FLOATS = T::TILE_FLOATS, BF_BUF_FLOATS = T::BUF_FLOATS, BF_EPI_FLOATS = T::EPI_FLOATS;
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

	// staging role: thread owns 16-byte slot `sch` of tile rows srow + BF_RPP * j; a wave instruction moves whole 128-byte lines
	const uint32_t sch = t & (BF_CH - 1), srow = t / BF_CH;
	const uint32_t nchunks = a.stride / 4;
	const floatx4 *qsrc[BF_PASSES], *xsrc[BF_PASSES_R];
#pragma unroll
	for (int j = 0; j < BF_PASSES; j++)
		qsrc[j] = reinterpret_cast<const floatx4 *>(a.queries + (size_t) min(q0 + srow + BF_RPP * j, a.nq - 1) * a.qstride);
#pragma unroll
	for (int j = 0; j < BF_PASSES_R; j++)
		xsrc[j] = reinterpret_cast<const floatx4 *>(a.vec + (size_t) min(r0 + srow + BF_RPP * j, a.n - 1) * a.stride);

	floatx16 acc[2][NJ];
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < NJ; j++)
#pragma unroll
			for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
	const uint32_t wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves over the block tile
	const uint32_t kk = lane >> 5, col = lane & 31;

	// What the epilogue compares with is fetched NOW, behind the K loop: every block of a launch takes the same time, so the blocks of
	// a CU reach their epilogues together, and an epilogue that loads its 32 bounds per lane one dependent L2 round trip after the
	// other (rounds 1-4) leaves the matrix pipe idle.  The tile's 128 bounds and |q|^2 go to LDS (read back four at a time), the two
	// |x|^2 of the lane's columns to registers.
	float *epi = bf_lds + (size_t) BF_NBUF * BF_BUF_FLOATS;
	// Rows that pass the filter (about twenty per block) are collected in LDS and appended to their queries' candidate lists by the
	// whole block at once: in rounds 1-4 every passing element was a returning global atomic behind its own `s_waitcnt vmcnt(0)` inside
	// a divergent branch — a handful of dependent L2 round trips per wave at the end of every block, with the matrix pipe idle.
	uint32_t *pass_cnt = reinterpret_cast<uint32_t *>(epi + BF_EPI_FLOATS);
	uint2 *pass_list = reinterpret_cast<uint2 *>(epi + BF_EPI_FLOATS + 4);
	if (t == 0) *pass_cnt = 0u;
	if (t < BF_TQ)
	{
		const uint32_t qi = min(q0 + (uint32_t) t, a.nq - 1);
		epi[t] = a.qbound[qi];
		epi[BF_TQ + t] = a.qnorm[qi];
	}
	float xs2[NJ];
#pragma unroll
	for (int j = 0; j < NJ; j++)
	{
		const float xn = a.xnorm[min(r0 + wn * (32 * NJ) + j * 32 + col, a.n - 1)];
		xs2[j] = (a.func == F_COSINE) ? __builtin_sqrtf(xn) : xn;
	}

	// Tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write).  The LDS image is what the hardware
	// writes — wave-uniform base + lane * 16: BF_RPI rows of whole lines per instruction, no padding — so bank conflicts are avoided on
	// the SOURCE side: the lane that fills slot p of row r fetches chunk p ^ swizzle(r).  No select on a loaded value either: the query
	// copy is zero padded to whole K steps, and a row chunk beyond the row's end re-reads the row's last chunk (times zero: nothing).
	const uint32_t gch = sch ^ bf_swz(srow);                    // (swizzle(srow + BF_RPP * j) == swizzle(srow): the step is a multiple of 16)
	auto fetch = [&](uint32_t ks, uint32_t buf)
	{
		const uint32_t c = ks * BF_CH + gch;
		const uint32_t cc = min(c, nchunks - 1);
		float *As = bf_lds + (size_t) buf * BF_BUF_FLOATS + (wave * BF_RPI) * BF_LS, *Bs = As + BF_TILE_FLOATS;
#pragma unroll
		for (int j = 0; j < BF_PASSES; j++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (qsrc[j] + c),
											 (__attribute__((address_space(3))) void *) (As + BF_RPP * j * BF_LS), 16, 0, 0);
#pragma unroll
		for (int j = 0; j < BF_PASSES_R; j++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (xsrc[j] + cc),
											 (__attribute__((address_space(3))) void *) (Bs + BF_RPP * j * BF_LS), 16, 0, 0);
	};
	// An MFMA may sum a dot product in any k order as long as both operands use the same one: lane (col, kk) takes the four k of ONE
	// ds_read_b128 (chunk 2g + kk of its row) and feeds them to four consecutive MFMAs — 16 ds_read_b128 per 64 MFMAs.
	const uint32_t swz = bf_swz(col);                           // (the tile rows of this lane are col + 32 * i + 64 * wm: same swizzle)
	uint32_t roff[BF_TK / 8];                                  // float offset of k-group g's slot in this lane's row
#pragma unroll
	for (int g = 0; g < BF_TK / 8; g++) roff[g] = ((2 * g + kk) ^ swz) * 4;
	auto contract = [&](uint32_t buf)
	{
		const float *As = bf_lds + (size_t) buf * BF_BUF_FLOATS + (wm * 64 + col) * BF_LS;
		const float *Bs = bf_lds + (size_t) buf * BF_BUF_FLOATS + BF_TILE_FLOATS + (wn * (32 * NJ) + col) * BF_LS;
#pragma unroll
		for (int g = 0; g < BF_TK / 8; g++)
		{
			const uint32_t o = roff[BF_ABLATE == 2 ? 0 : g];
			floatx4 av[2], bv[NJ];
#pragma unroll
			for (int i = 0; i < 2; i++) av[i] = *reinterpret_cast<const floatx4 *>(As + i * 32 * BF_LS + o);
#pragma unroll
			for (int j = 0; j < NJ; j++) bv[j] = *reinterpret_cast<const floatx4 *>(Bs + j * 32 * BF_LS + o);
#pragma unroll
			for (int c = 0; c < 4; c++)
			{
#pragma unroll
				for (int i = 0; i < 2; i++)
#pragma unroll
					for (int j = 0; j < NJ; j++)
						acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
			}
		}
	};

	unsigned long long c0 = 0, r0c = 0;
	if (a.clocks) { c0 = __builtin_readcyclecounter(); r0c = wall_clock64(); }
	// Software pipeline over two LDS buffers: the loads of K step ks + 1 are issued before the MFMAs of step ks, into the other buffer;
	// one barrier per step.  (Measured and NOT kept, profiles/r5e_mfma_tile_variants.txt: a 64-float K step with one buffer 129 TFLOP/s,
	// 64 x 128 wave tiles 128-130, operand reads written out one k-group ahead with counted lgkmcnt waits and the barrier in front of
	// the last k-group — no LDS or memory round trip exposed inside the loop — 129; three buffers with a counted vmcnt 132.5; the next
	// step's loads issued between the MFMAs 130 / 124; s_setprio around the MFMAs 131.7; a pipeline across tiles in resident blocks +0.6 % /
	// -1.5 % (profiles/r5y_*); this plain form 134-136 with 128 x 128 tiles, 138 with 256 x 256.)
	fetch(0, 0);
	// The tile loads are LDS-bound DMA (global_load_lds): what orders them before the other waves' LDS reads is vmcnt reaching 0 on the
	// ISSUING wave before it arrives at the barrier.  hipcc emits that wait today, but nothing obliges it to (gfx950 has back-off
	// barriers: no automatic waitcnt; a workgroup fence does not wait on vmcnt) — a compiler that dropped it would let a wave read tile
	// rows another wave's DMA has not landed yet, and the filter would lose true neighbours silently.  Said explicitly (vmcnt(0) only:
	// expcnt / lgkmcnt left alone); free where the compiler already waits.
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	for (uint32_t ks = 0; ks < a.ksteps; ks++)
	{
#if BF_ABLATE != 3                                              // (ablation builds — wrong answers, timing only: profiles/r5l_mfma_ablation.txt)
		fetch(min(ks + 1, a.ksteps - 1), (ks + 1) & 1);            // its readers passed the previous barrier (branch-free: the last step re-reads itself)
#endif
		__builtin_amdgcn_sched_barrier(0);
		contract(ks & 1);
#if BF_ABLATE == 1
		__builtin_amdgcn_s_waitcnt(0);                             // the loads are still waited for; only the rendezvous of the four waves is gone
#else
		__builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): this wave's DMA of step ks + 1 has landed before anybody reads it
		__syncthreads();
#endif
	}
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
			for (int j = 0; j < NJ; j++)
			{
				const uint32_t r = r0 + wn * (32 * NJ) + j * 32 + col;
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
					if (BF_ABLATE) pass = pass && dot == 12345.678f;      // (timing-only builds compute garbage: keep it out of the lists)
					if (pass && rok && q < a.nq)
					{
						const uint32_t slot = atomicAdd(pass_cnt, 1u);      // (LDS)
						if (slot < (uint32_t) BF_PASS_CAP) pass_list[slot] = make_uint2(q, r);
						else                                              // a block with more passes than the list holds: straight to the global list
						{
							const uint32_t pos = atomicAdd(&a.cand_cnt[q], 1u);
							if (pos < a.cap) a.cand[(size_t) q * a.cap + pos] = r;
						}
					}
				}
			}
		}
	__syncthreads();
	const uint32_t npass = min(*pass_cnt, (uint32_t) BF_PASS_CAP);
	for (uint32_t i = (uint32_t) t; i < npass; i += BF_THREADS)
	{
		const uint2 e = pass_list[i];
		const uint32_t pos = atomicAdd(&a.cand_cnt[e.x], 1u);
		if (pos < a.cap) a.cand[(size_t) e.x * a.cap + pos] = e.y;
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
