// device_build.h — insert path on the device mirror: bindPoint (hnswalg.cpp:225-232) =
// searchBaseLayer(ef = efConstruction) + mutuallyConnectNewElement (hnswalg.cpp:155-223)
// with getNeighborsByHeuristic (hnswalg.cpp:117-153).
//
// A batch of already-stored, un-linked elements [first, first+count) is linked in three
// device steps (driver: hnsw_gpu_index_link in hnsw_gpu.hip):
//   1. the fused search kernel (device_search.h, mode 1) with the batch rows as queries;
//   2. select_links_kernel  — one wavefront per new element: heuristic choice of <= M
//      neighbours, its own link list written, one (target, new) pair emitted per link;
//   3. pairs sorted by (target, new) and reverse_links_kernel — one wavefront per distinct
//      target applies its incoming links IN ORDER: append while the list has room,
//      otherwise re-select maxM of {old links + new} by the same heuristic.
// With batch size 1 this is exactly the reference's serial insert (same candidate order,
// same pair comparisons, same link order) and the graph is bit-identical to the oracle's;
// larger batches trade that for parallelism (members of one batch do not see each other
// in step 1) and are flagged as a different — but equally valid — graph.
#pragma once
#include "device_search.h"

namespace pgemb {

constexpr int BUILD_KB = 4;     // chunk-steps per load batch in the builder kernels (any dim)

struct BuildArgs
{
	const float *vec;
	uint32_t    *links;
	uint32_t dim, stride, nchunks, kiters, qpad_floats, maxM, M, lstride;
	uint32_t first, count;              // new elements
	const uint32_t *cand_idx;           // [count][efc] ascending by (dist, idx)
	const float    *cand_dist;
	const uint32_t *cand_cnt;
	uint32_t efc;
	uint64_t *pairs;                    // out: (target << 32) | new
	uint32_t *npairs;                   // atomic cursor
	const uint64_t *sorted_pairs;       // step 3 input
	uint32_t pair_slots;
	const uint32_t *seg_start;
	const uint32_t *nseg;
	uint32_t *ticket;
	uint32_t wave_bytes;                // LDS per wave
	uint32_t single;                    // count == 1 (the reference's serial insert): the selected neighbours are distinct targets, so the
	uint32_t *seg_out;                  // pairs need no sort — select_links_kernel writes one segment per pair (seg_out, nseg_out) and ends
	uint32_t *nseg_out;                 // the pair array itself; reverse_links_kernel then reads `pairs` as its sorted input
};

// Copy one padded row into the LDS query image (zero tail up to qpad_floats).
__device__ __forceinline__ void stage_row(float *qf, const float *row, uint32_t stride, uint32_t qpad_floats, int lane)
{
	for (uint32_t e = lane; e < qpad_floats; e += 64)
	{
		const float t = row[e < stride ? e : stride - 1];        // unconditional load, then select
		qf[e] = (e < stride) ? t : 0.f;
	}
	wave_sync();
}

// Ascending rank sort of n 64-bit keys (unique) from `in` to `out`; wave-cooperative.
__device__ __forceinline__ void rank_sort(const uint64_t *in, uint64_t *out, uint32_t n, bool descending, int lane)
{
	for (uint32_t b = 0; b < n; b += 64)
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
	wave_sync();
}

// getNeighborsByHeuristic, hnswalg.cpp:130-150, over candidates already in the reference's
// pop order (closest first, equal distances by larger idx first: the max-heap on
// (-dist, idx), :125-128).  ckey[k] = ord(dist)<<32 | ~idx.  Selected elements are left in
// sel_key[] as ord(dist)<<32 | idx.  Returns how many were selected (<= NN).
template <int FUNC>
__device__ __forceinline__ uint32_t heuristic_select(const BuildArgs &a, float *qf, const uint64_t *ckey, uint32_t ncand,
													 uint32_t NN, uint64_t *sel_key, uint32_t *sel_id, float *tmpd, int lane)
{
	const float4 *q4 = reinterpret_cast<const float4 *>(qf);
	uint32_t nsel = 0;
	for (uint32_t k = 0; k < ncand && nsel < NN; k++)                // :130-132
	{
		const uint64_t key = ckey[k];
		const uint32_t c = ~(uint32_t) key;
		const float dist_to_query = unord_f32((uint32_t) (key >> 32));
		bool good = true;
		if (nsel > 0)                                                 // :137-148
		{
			stage_row(qf, a.vec + (size_t) c * a.stride, a.stride, a.qpad_floats, lane);
			float qnorm = 0.f;
			if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);
			const uint32_t *ids = sel_id;
			bool closer = false;
			for (uint32_t b = 0; b < nsel; b += 64)                   // 64 selected rows per pass
			{
				const uint32_t nb = nsel - b < 64 ? nsel - b : 64;
				auto by_id = [ids, b](uint32_t r) { return ids[b + r]; };
				score_rows<FUNC, BUILD_KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nb, tmpd, lane);
				wave_sync();
				const float curdist = finish_dist<FUNC>(tmpd[lane], tmpd[OUT2 + lane], qnorm);
				closer |= ((uint32_t) lane < nb) && (curdist < dist_to_query);   // curdist < dist_to_query, :143
				wave_sync();
			}
			good = __ballot(closer) == 0;
		}
		if (good)                                                     // :149
		{
			if (lane == 0)
			{
				sel_id[nsel] = c;
				sel_key[nsel] = (key & 0xFFFFFFFF00000000ull) | c;
			}
			nsel++;
			wave_sync();
		}
	}
	return nsel;
}

// LDS carve shared by both kernels (per wave): query image | keyA[cap] | keyB[cap] | ids[cap] |
// dist[cap] | tmpd[cap]   with cap = max(efc, maxM + 1) rounded up to 8.
__device__ __forceinline__ uint32_t build_cap(const BuildArgs &a)
{
	uint32_t c = a.efc > a.maxM + 1 ? a.efc : a.maxM + 1;
	if (c < 128) c = 128;               // tmpd doubles as a score_rows output (2 x 64 sums)
	return (c + 7) & ~7u;
}

// Step 2: one wavefront per new element.
template <int FUNC>
__global__ __launch_bounds__(256) void select_links_kernel(const BuildArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const uint32_t w = blockIdx.x * (blockDim.x >> 6) + wib;
	if (w >= a.count) return;
	const uint32_t cap = build_cap(a);
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float    *qf   = reinterpret_cast<float *>(my);
	uint64_t *keyA = reinterpret_cast<uint64_t *>(my + (size_t) a.qpad_floats * 4);
	uint64_t *keyB = keyA + cap;
	uint32_t *ids  = reinterpret_cast<uint32_t *>(keyB + cap);
	float    *tmpd = reinterpret_cast<float *>(ids + cap) + cap;

	const uint32_t p = a.first + w;
	const uint32_t ncand = a.cand_cnt[w];
	const uint32_t *ci = a.cand_idx + (size_t) w * a.efc;
	const float *cd = a.cand_dist + (size_t) w * a.efc;

	// candidates into pop order of the (-dist, idx) heap (hnswalg.cpp:125-128)
	for (uint32_t i = lane; i < ncand; i += 64)
		keyA[i] = ((uint64_t) ord_f32(cd[i]) << 32) | (uint32_t) ~ci[i];
	wave_sync();
	rank_sort(keyA, keyB, ncand, false, lane);

	uint32_t nsel;
	if (ncand < a.M)                                     // hnswalg.cpp:119-120: keep them all
	{
		for (uint32_t i = lane; i < ncand; i += 64)
			keyA[i] = (keyB[i] & 0xFFFFFFFF00000000ull) | (uint32_t) ~(uint32_t) keyB[i];
		nsel = ncand;
		wave_sync();
	}
	else
		nsel = heuristic_select<FUNC>(a, qf, keyB, ncand, a.M, keyA, ids, tmpd, lane);

	// own link list = selected, farthest first ((dist, idx) max-heap pops, hnswalg.cpp:164-181)
	rank_sort(keyA, keyB, nsel, true, lane);
	uint32_t *mine = a.links + (size_t) p * a.lstride;
	for (uint32_t j = lane; j < a.lstride; j += 64)
		mine[j] = (j < nsel) ? (uint32_t) keyB[j] : LINK_NONE;

	// one reverse edge per link, applied in step 3 in the same order (hnswalg.cpp:183)
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(a.npairs, nsel);
	base = __builtin_amdgcn_readfirstlane(base);
	for (uint32_t j = lane; j < nsel; j += 64)
		a.pairs[base + j] = ((uint64_t) (uint32_t) keyB[j] << 32) | p;
	if (a.single)                                       // one new element: every pair is a segment of its own, in link order (:183)
	{
		for (uint32_t j = lane; j < nsel; j += 64) a.seg_out[j] = j;
		if (lane == 0)
		{
			*a.nseg_out = nsel;
			if (nsel < a.pair_slots) a.pairs[nsel] = ~0ull;     // end of the pair array
		}
	}
}

// Step 3 helper: mark the first pair of every target.
__global__ void mark_segments_kernel(const uint64_t *__restrict__ sorted, uint32_t slots, uint32_t *seg_start, uint32_t *nseg)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= slots) return;
	const uint64_t k = sorted[i];
	if (k == ~0ull) return;
	if (i == 0 || (uint32_t) (sorted[i - 1] >> 32) != (uint32_t) (k >> 32))
		seg_start[atomicAdd(nseg, 1u)] = i;
}

// Step 3: one wavefront per target element, incoming links applied in order.
template <int FUNC>
__global__ __launch_bounds__(256) void reverse_links_kernel(const BuildArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const uint32_t cap = build_cap(a);
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float    *qf   = reinterpret_cast<float *>(my);
	const float4 *q4 = reinterpret_cast<const float4 *>(my);
	uint64_t *keyA = reinterpret_cast<uint64_t *>(my + (size_t) a.qpad_floats * 4);
	uint64_t *keyB = keyA + cap;
	uint32_t *ids  = reinterpret_cast<uint32_t *>(keyB + cap);
	float    *cdist = reinterpret_cast<float *>(ids + cap);
	float    *tmpd = cdist + cap;
	// current list lives in LDS behind everything else: cur[0] = incoming, cur[1..cnt] = links
	uint32_t *cur  = reinterpret_cast<uint32_t *>(tmpd + cap);
	const uint32_t nseg = *a.nseg;

	for (;;)
	{
		uint32_t s = 0;
		if (lane == 0) s = atomicAdd(a.ticket, 1u);
		s = __builtin_amdgcn_readfirstlane(s);
		if (s >= nseg) break;
		uint32_t i = a.seg_start[s];
		const uint32_t t = (uint32_t) (a.sorted_pairs[i] >> 32);
		uint32_t *list = a.links + (size_t) t * a.lstride;

		// load + compact the current list (an imported image may have holes)
		uint32_t cnt = 0;
		for (uint32_t j0 = 0; j0 < a.lstride; j0 += 64)
		{
			const uint32_t j = j0 + lane;
			const uint32_t v = (j < a.lstride) ? list[j] : LINK_NONE;
			const uint64_t m = __ballot(v != LINK_NONE);
			if (v != LINK_NONE) cur[1 + cnt + lane_rank(m)] = v;
			cnt += (uint32_t) __builtin_popcountll(m);
		}
		wave_sync();

		for (; i < a.pair_slots; i++)
		{
			const uint64_t pk = a.sorted_pairs[i];
			if ((uint32_t) (pk >> 32) != t || pk == ~0ull) break;
			const uint32_t p = (uint32_t) pk;
			if (cnt < a.maxM)                                    // hnswalg.cpp:194-196
			{
				if (lane == 0) cur[1 + cnt] = p;
				cnt++;
				wave_sync();
				continue;
			}
			// hnswalg.cpp:197-220: re-select maxM of {new, old links} around t
			if (lane == 0) cur[0] = p;
			stage_row(qf, a.vec + (size_t) t * a.stride, a.stride, a.qpad_floats, lane);
			float qnorm = 0.f;
			if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);
			for (uint32_t b = 0; b <= cnt; b += 64)
			{
				const uint32_t nb = cnt + 1 - b < 64 ? cnt + 1 - b : 64;
				const uint32_t *cc = cur;
				auto by_id = [cc, b](uint32_t r) { return cc[b + r]; };
				score_rows<FUNC, BUILD_KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nb, tmpd, lane);
				wave_sync();
				const float dl = finish_dist<FUNC>(tmpd[lane], tmpd[OUT2 + lane], qnorm);
				if ((uint32_t) lane < nb) keyA[b + lane] = ((uint64_t) ord_f32(dl) << 32) | (uint32_t) ~cur[b + lane];
				wave_sync();
			}
			rank_sort(keyA, keyB, cnt + 1, false, lane);         // pop order of (-dist, idx)
			const uint32_t nsel = heuristic_select<FUNC>(a, qf, keyB, cnt + 1, a.maxM, keyA, ids, tmpd, lane);
			rank_sort(keyA, keyB, nsel, true, lane);             // :214-219: farthest first
			for (uint32_t j = lane; j < nsel; j += 64) cur[1 + j] = (uint32_t) keyB[j];
			cnt = nsel;
			wave_sync();
		}
		for (uint32_t j = lane; j < a.lstride; j += 64)
			list[j] = (j < cnt) ? cur[1 + j] : LINK_NONE;
	}
}

}  // namespace pgemb
