// device_insert.h — the reference's serial insert of ONE element (hnswalg.cpp:225-232, 155-223, 117-153) as two launches
// built for LATENCY (hnsw_gpu_index_insert_one / _insert_candidates; the batched builder of device_build.h is built for
// throughput and stays as it is).
//
// Where a serial insert's time went (profiles/r3h_serial_insert.txt): getNeighborsByHeuristic is a chain — candidate k is
// compared with the neighbours chosen before it (hnswalg.cpp:137-148), so one wavefront walking the candidates pays a row
// fetch and up to M/4 scoring passes per candidate, 64 candidates in a row, and each of the <= M neighbours whose list is
// full repeats that over maxM + 1 candidates.  But the only data-dependent part of the chain is WHICH earlier candidates
// were chosen: every comparison it can ask for is  dist(candidate k, candidate j) < dist(candidate k, query)  with j < k in
// pop order — one BIT per pair.  So:
//   step 1  insert_select_kernel   the strict lower triangle of that bit matrix is cut into units of (candidate k, 16 earlier
//           candidates) and every wavefront of the grid takes one: stage row k, score 16 rows, one ballot, 16 bits stored.
//           Two memory round trips, whatever efConstruction is.  The block that finishes last gathers the bits into LDS
//           (candidates x candidates / 8 bytes) and runs the chain on them: "does row k's mask meet the chosen set" is one
//           AND per 64 candidates, the next row's mask is already on its way while this one is tested.  It writes the new
//           element's link list and the list of targets.  Block 0 also stores the row (the append).
//   step 2  insert_reverse_kernel  one block per target (hnswalg.cpp:183-222): room in the list -> append; full -> the same
//           units over {new, old links} around the target across the block's wavefronts, chain by wave 0.  Each block
//           writes its target's list into the mirror AND into the caller's pinned staging; the block that finishes last
//           stores the completion flag.
// The distances are the ones device_build.h computes (same staged "query", same rows, per-row summation order independent
// of the pass shape), so the graph stays byte-identical to the oracle's: tests/test_gpu_build.py, tests/emu (insert scenario).
#pragma once
#include "device_build.h"

namespace pgemb {

struct InsertArgs
{
	BuildArgs b;                  // vec, links, dims, M, maxM, lstride, efc, cand_* (one element), first = the new element
	const float    *src_row;      // the point, dim floats (pinned host memory); NULL = the row is stored already
	const uint64_t *src_label;
	uint64_t *labels;
	uint32_t *targets;            // step 1 -> step 2: the chosen neighbours in link order
	uint32_t *ntargets;
	uint16_t *bits;               // step 1: the bit matrix in device memory, [side][side / 16] pieces of 16 bits
	uint32_t *lists_out;          // [(maxM + 1)][lstride]: row 0 = the new element's list, row 1 + j = the list of link slot j
	uint32_t *done1, *done2;      // block counters (zero between calls: the last block resets them)
	uint32_t *flag;               // completion flag in pinned host memory
	uint32_t nw;                  // wavefronts per block, step 1
	uint32_t nw2;                 // wavefronts per block, step 2 (one block per target: more wavefronts = fewer rounds of units)
	uint32_t side;                // candidates the bit matrix has room for (a multiple of 64)
	uint32_t cap;                 // key array length
	uint32_t bind;                // 0 = element 0: stored, never bound (hnswalg.cpp:228)
	uint32_t ncand_p1;            // 1 + the number of candidates when the host knows it (no read over the bus for it), else 0
};

constexpr int INS_KB = 4, INS_RPG = 4;       // one scoring pass = 16 rows = one unit, 4 chunk-steps per load batch
constexpr uint32_t INS_MAX_SIDE = 512;      // the chain keeps the chosen set as one 64-bit word per lane-of-eight

// Block-shared part of the LDS carve; the per-wave parts (query image, 2 x 64 sums) follow it.
struct InsertLds
{
	uint64_t *pop, *keyA, *keyB;  // [cap] each: pop order | scratch | sorted output
	uint32_t *cur;                // [maxM + 2]: cur[0] = incoming, cur[1..] = the target's links
	uint32_t *sh;                 // [4] block-shared words
	uint64_t *m64;                // [side][side / 64]: bit j of row k = candidate j is closer to candidate k than the query is
	float    *qf;                 // this wave's query image
	float    *tmpd;               // this wave's 2 x 64 sums
};

__device__ __forceinline__ InsertLds insert_carve(const InsertArgs &a, unsigned char *smem, uint32_t wib)
{
	InsertLds L;
	L.pop = reinterpret_cast<uint64_t *>(smem);
	L.keyA = L.pop + a.cap;
	L.keyB = L.keyA + a.cap;
	L.m64 = L.keyB + a.cap;
	L.cur = reinterpret_cast<uint32_t *>(L.m64 + (size_t) a.side * (a.side / 64));
	L.sh = L.cur + ((a.b.maxM + 2 + 3) & ~3u);
	float *waves = reinterpret_cast<float *>(L.sh + 4);
	L.qf = waves + (size_t) wib * (a.b.qpad_floats + 128);
	L.tmpd = L.qf + a.b.qpad_floats;
	return L;
}

// rank_sort of device_build.h across a block: wavefront w ranks keys w*64.., (w + W)*64.. .  Callers put a block barrier behind it.
__device__ __forceinline__ void rank_sort_block(const uint64_t *in, uint64_t *out, uint32_t n, bool descending, uint32_t w, uint32_t W, int lane)
{
	for (uint32_t b = w * 64; b < n; b += W * 64)
	{
		const uint32_t i = b + lane;
		if (i < n)
		{
			const uint64_t k = in[i];
			uint32_t r = 0;
			for (uint32_t j = 0; j < n; j++) r += (descending ? in[j] > k : in[j] < k) ? 1u : 0u;
			out[r] = k;
		}
	}
}

// Unit u of the triangle -> (candidate k >= 1, group g of 16 earlier candidates, 16 * g < k).  Candidates 16i+1 .. 16i+16 have
// i + 1 groups each, so band i holds 16 (i + 1) units and 8 i (i + 1) units lie before it.
__device__ __forceinline__ void unit_of(uint32_t u, uint32_t &k, uint32_t &g)
{
	uint32_t i = 0;
	while (8u * (i + 1u) * (i + 2u) <= u) i++;
	const uint32_t r = u - 8u * i * (i + 1u);
	k = 16u * i + 1u + r / (i + 1u);
	g = r % (i + 1u);
}
__host__ __device__ __forceinline__ uint32_t units_for(uint32_t ncand)      // units that cover candidates 1 .. ncand - 1 (whole bands)
{
	const uint32_t bands = ncand > 1 ? (ncand - 1 + 15) / 16 : 0;
	return 8u * bands * (bands + 1u);
}

// One unit: bits[k][g] = for the 16 candidates j = 16g .. 16g + 15 (j < k, pop order): dist(candidate k staged as the query,
// row of candidate j) < dist(candidate k, the insert's query) — hnswalg.cpp:141-146 for every pair the chain can ask about.
template <int FUNC, typename Bits>
__device__ __forceinline__ void triangle_unit(const BuildArgs &a, const uint64_t *pop, uint32_t k, uint32_t g, Bits *bits, uint32_t pieces,
											  float *qf, float *tmpd, int lane)
{
	const float4 *q4 = reinterpret_cast<const float4 *>(qf);
	const uint64_t key = pop[k];
	const float dist_to_query = unord_f32((uint32_t) (key >> 32));
	stage_row(qf, a.vec + (size_t) (~(uint32_t) key) * a.stride, a.stride, a.qpad_floats, lane);
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);
	const uint32_t b = 16u * g;
	const uint32_t nb = k - b < 16u ? k - b : 16u;
	auto by_pos = [pop, b](uint32_t r) { return ~(uint32_t) pop[b + r]; };
	score_rows_fit<FUNC, INS_KB, INS_RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_pos, nb, tmpd, lane);
	wave_sync();
	const float d = finish_dist<FUNC>(tmpd[lane], tmpd[OUT2 + lane], qnorm);
	const uint64_t m = __ballot((uint32_t) lane < nb && d < dist_to_query);
	if (lane == 0) bits[(size_t) k * pieces + g] = (Bits) m;
	wave_sync();
}

// The chain of getNeighborsByHeuristic (hnswalg.cpp:130-150) over the finished bit matrix; one wavefront.  Lane w < W keeps
// word w of the chosen set; row k + 1's words are loaded before row k is decided (the loads do not depend on the decisions).
// Leaves the chosen candidates in keyA as ord(dist) << 32 | idx and returns how many.
__device__ __forceinline__ uint32_t chain_select(const uint64_t *pop, uint32_t ncand, uint32_t NN, const uint64_t *m64, uint32_t W,
												 uint64_t *keyA, int lane)
{
	uint32_t nsel = 0;
	uint64_t chosen = 0;
	const bool mine = (uint32_t) lane < W;
	uint64_t row = (mine && ncand) ? m64[lane] : 0ull;
	for (uint32_t k = 0; k < ncand && nsel < NN; k++)               // :130-132
	{
		const uint64_t nxt = (mine && k + 1 < ncand) ? m64[(size_t) (k + 1) * W + lane] : 0ull;
		if (__ballot((row & chosen) != 0ull) == 0)                   // :137-149: nobody chosen so far is closer to it than the query
		{
			if ((uint32_t) lane == (k >> 6)) chosen |= 1ull << (k & 63);
			if (lane == 0)
			{
				const uint64_t key = pop[k];
				keyA[nsel] = (key & 0xFFFFFFFF00000000ull) | (uint32_t) ~(uint32_t) key;
			}
			nsel++;
		}
		row = nxt;
	}
	wave_sync();
	return nsel;
}

// Block counter: true in every thread of the block that arrives last.  Everything the block wrote is released first (at
// `system` scope when the data is for the host), and the last block acquires what the others wrote.
template <bool SYSTEM>
__device__ __forceinline__ bool last_block(uint32_t *ctr, uint32_t *sh_word)
{
	if (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
	else        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	__syncthreads();
	if (threadIdx.x == 0) *sh_word = atomicAdd(ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	const bool last = *sh_word != 0;
	if (last)
	{
		if (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
		else        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
	}
	return last;
}

// Step 1.
template <int FUNC>
__global__ __launch_bounds__(512) void insert_select_kernel(const InsertArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const InsertLds L = insert_carve(a, smem, wib);
	const uint32_t p = a.b.first;
	const uint32_t W = a.side / 64, pieces = a.side / 16;

	if (a.src_row && blockIdx.x == 0)                              // the append: padded row and label (links: below, by the last block)
	{
		for (uint32_t c = threadIdx.x; c < a.b.stride; c += blockDim.x)
			const_cast<float *>(a.b.vec)[(size_t) p * a.b.stride + c] = (c < a.b.dim) ? a.src_row[c] : 0.f;
		if (threadIdx.x == 0) a.labels[p] = a.src_label ? *a.src_label : (uint64_t) p;
	}

	const uint32_t ncand = !a.bind ? 0u : (a.ncand_p1 ? a.ncand_p1 - 1u : a.b.cand_cnt[0]);
	const bool chain = ncand >= a.b.M;                              // hnswalg.cpp:119-120: fewer than M candidates are all kept
	// candidates into pop order of the (-dist, idx) heap (:125-128) — every block for itself: it names the rows of its units
	for (uint32_t i = threadIdx.x; i < ncand; i += blockDim.x)
		L.keyA[i] = ((uint64_t) ord_f32(a.b.cand_dist[i]) << 32) | (uint32_t) ~a.b.cand_idx[i];
	__syncthreads();
	rank_sort_block(L.keyA, L.pop, ncand, false, wib, a.nw, lane);
	__syncthreads();
	if (chain)
		for (uint32_t u = blockIdx.x * a.nw + wib; u < units_for(ncand); u += gridDim.x * a.nw)
		{
			uint32_t k, g;
			unit_of(u, k, g);
			if (k < ncand) triangle_unit<FUNC>(a.b, L.pop, k, g, a.bits, pieces, L.qf, L.tmpd, lane);
		}
	if (!last_block<false>(a.done1, L.sh)) return;

	if (chain)                                                      // the bits into LDS, 16 at a time; what no unit wrote (j >= k) is zero
	{
		uint16_t *m16 = reinterpret_cast<uint16_t *>(L.m64);
		for (uint32_t i = threadIdx.x; i < ncand * pieces; i += blockDim.x)
		{
			const uint32_t k = i / pieces, g = i % pieces;
			m16[i] = 16u * g < k ? a.bits[i] : (uint16_t) 0;
		}
	}
	__syncthreads();
	if (wib == 0)
	{
		uint32_t nsel;
		if (chain)
			nsel = chain_select(L.pop, ncand, a.b.M, L.m64, W, L.keyA, lane);
		else
		{
			for (uint32_t i = lane; i < ncand; i += 64)
				L.keyA[i] = (L.pop[i] & 0xFFFFFFFF00000000ull) | (uint32_t) ~(uint32_t) L.pop[i];
			nsel = ncand;
			wave_sync();
		}
		// own link list = chosen, farthest first ((dist, idx) max-heap pops, hnswalg.cpp:164-181); one target per link (:183)
		rank_sort(L.keyA, L.keyB, nsel, true, lane);
		uint32_t *mine = a.b.links + (size_t) p * a.b.lstride;
		for (uint32_t j = lane; j < a.b.lstride; j += 64)
			mine[j] = (j < nsel) ? (uint32_t) L.keyB[j] : LINK_NONE;
		for (uint32_t j = lane; j < nsel; j += 64) a.targets[j] = (uint32_t) L.keyB[j];
		if (lane == 0)
		{
			*a.ntargets = nsel;
			atomicExch(a.done1, 0u);                                // ready for the next insert
		}
	}
}

// Step 2: block s = target s (blocks past the number of targets only count themselves done).
// (12 wavefronts: the scoring pass of 16 rows wants ~150-180 VGPRs, which a block of 16 wavefronts cannot have)
template <int FUNC>
__global__ __launch_bounds__(768) void insert_reverse_kernel(const InsertArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const InsertLds L = insert_carve(a, smem, wib);
	const float4 *q4 = reinterpret_cast<const float4 *>(L.qf);
	const uint32_t p = a.b.first, s = blockIdx.x;
	const uint32_t nt = *a.ntargets;
	const uint32_t W = a.side / 64, pieces = a.side / 16;

	if (s == 0)                                                     // the new element's own list, as step 1 left it in the mirror
		for (uint32_t j = threadIdx.x; j < a.b.lstride; j += blockDim.x)
			a.lists_out[j] = a.b.links[(size_t) p * a.b.lstride + j];
	if (s < nt)
	{
		const uint32_t t = a.targets[s];
		uint32_t *list = a.b.links + (size_t) t * a.b.lstride;
		if (wib == 0)
		{
			// load + compact the current list (an imported image may have holes)
			uint32_t cnt = 0;
			for (uint32_t j0 = 0; j0 < a.b.lstride; j0 += 64)
			{
				const uint32_t j = j0 + lane;
				const uint32_t v = (j < a.b.lstride) ? list[j] : LINK_NONE;
				const uint64_t m = __ballot(v != LINK_NONE);
				if (v != LINK_NONE) L.cur[1 + cnt + lane_rank(m)] = v;
				cnt += (uint32_t) __builtin_popcountll(m);
			}
			if (lane == 0)
			{
				if (cnt < a.b.maxM) { L.cur[1 + cnt] = p; L.sh[1] = cnt + 1; L.sh[2] = 0; }      // hnswalg.cpp:194-196
				else                { L.cur[0] = p;       L.sh[1] = cnt;     L.sh[2] = 1; }      // :197-220: re-select maxM of {new, old links} around t
			}
		}
		__syncthreads();
		uint32_t cnt = L.sh[1];
		if (L.sh[2])                                                // block-uniform
		{
			const uint32_t ncand = cnt + 1;
			// distances of the candidates around t, 16 per wavefront (t staged by each wavefront that has a group)
			for (uint32_t b = 16u * wib; b < ncand; b += 16u * a.nw2)
			{
				stage_row(L.qf, a.b.vec + (size_t) t * a.b.stride, a.b.stride, a.b.qpad_floats, lane);
				float qnorm = 0.f;
				if (FUNC == F_COSINE) qnorm = query_norm(q4, a.b.nchunks, a.b.kiters, lane);
				const uint32_t nb = ncand - b < 16u ? ncand - b : 16u;
				const uint32_t *cc = L.cur;
				auto by_id = [cc, b](uint32_t r) { return cc[b + r]; };
				score_rows_fit<FUNC, INS_KB, INS_RPG>(a.b.vec, a.b.stride, q4, a.b.nchunks, a.b.kiters, by_id, nb, L.tmpd, lane);
				wave_sync();
				const float dl = finish_dist<FUNC>(L.tmpd[lane], L.tmpd[OUT2 + lane], qnorm);
				if ((uint32_t) lane < nb) L.keyA[b + lane] = ((uint64_t) ord_f32(dl) << 32) | (uint32_t) ~L.cur[b + lane];
				wave_sync();
			}
			__syncthreads();
			rank_sort_block(L.keyA, L.pop, ncand, false, wib, a.nw2, lane);     // pop order of (-dist, idx)
			for (uint32_t i = threadIdx.x; i < ncand * W; i += blockDim.x) L.m64[i] = 0ull;
			__syncthreads();
			uint16_t *m16 = reinterpret_cast<uint16_t *>(L.m64);
			for (uint32_t u = wib; u < units_for(ncand); u += a.nw2)
			{
				uint32_t k, g;
				unit_of(u, k, g);
				if (k < ncand) triangle_unit<FUNC>(a.b, L.pop, k, g, m16, pieces, L.qf, L.tmpd, lane);
			}
			__syncthreads();
			if (wib == 0)
			{
				const uint32_t nsel = chain_select(L.pop, ncand, a.b.maxM, L.m64, W, L.keyA, lane);
				rank_sort(L.keyA, L.keyB, nsel, true, lane);         // :214-219: farthest first
				for (uint32_t j = lane; j < nsel; j += 64) L.cur[1 + j] = (uint32_t) L.keyB[j];
				cnt = nsel;
				wave_sync();
			}
		}
		if (wib == 0)
			for (uint32_t j = lane; j < a.b.lstride; j += 64)
			{
				const uint32_t v = (j < cnt) ? L.cur[1 + j] : LINK_NONE;
				list[j] = v;
				a.lists_out[(size_t) (1 + s) * a.b.lstride + j] = v;
			}
	}
	if (last_block<true>(a.done2, L.sh) && threadIdx.x == 0)
	{
		atomicExch(a.done2, 0u);
		__hip_atomic_store(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

}  // namespace pgemb
