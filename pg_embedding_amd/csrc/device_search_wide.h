// device_search_wide.h — searchBaseLayer / searchKnn (hnswalg.cpp:42-114, 234-252, 256-277) for beams of any width.
//
// The reference's scan doubles efSearch until a search comes back short (embedding.c:329-343), so a scan that consumes a
// whole index of n rows ends with beams of n/2 and n (a beam wider than the index is clamped to it: the same walk).  The
// generic form of device_search.h keeps its two sets as unsorted arrays and finds every extreme by a full scan: O(ef) per pop
// and per eviction, O(ef^2) for the output order — fine up to a few thousand, minutes at 200 000.  This form keeps the same
// two arrays (per-slot HBM scratch, L1-bypassing accesses) with a second level:
//
//   candidates : cand[0..csize) keys ord(dist)<<32 | ~idx, in chunks of CH keys; cmin[c] = smallest key of chunk c (LDS).
//                pop = min over cmin (LDS scan) + one chunk scan for the position; the last key moves into the hole and the two
//                chunks touched get their minimum recomputed: O(ef/CH + CH) per pop.  When the array is full (capacity 2*ef;
//                only possible once the result set is full) the dead candidates — distance above the bound: they can never be
//                expanded, the bound only shrinks — are dropped in one in-place pass: amortised O(1) per accepted element.
//   results    : res[0..rsize) keys ord(dist)<<32 | idx, same chunks; rmax[c] = largest key of chunk c (LDS).  Evicting the
//                largest = max over rmax + one chunk scan, overwrite, recompute that chunk's maximum; lowerBound = max over rmax.
//   output     : searchKnn's order (dist, label) — or (dist, idx) for the base-layer form — by a bitonic sort of (key, label)
//                pairs in the slot's scratch (the candidate array is dead by then): O(n log^2 n / 64) wave steps.
//
// Every decision is the reference's, on the same keys as the other forms (tie-breaks included), so ids, distance bits, E_q, H_q
// and the pop sequence equal the oracle's (tests/test_gpu_search.py::test_wide_beams_…).  One wavefront per query as everywhere;
// a walk with a beam of 200 000 is a chain of 200 000 hops whatever the container — this form makes each of them cost
// microseconds, not the better part of a millisecond.
#pragma once
#include "device_search.h"

namespace pgemb {

// recompute the extreme of chunk c of array A (n valid keys in all) into ext[c]; empty chunk -> the identity
template <bool MIN>
__device__ __forceinline__ void wide_refresh(const uint64_t *A, uint32_t n, uint32_t c, uint32_t CH, uint64_t *ext, int lane)
{
	const uint32_t base = c * CH;
	uint64_t v = MIN ? ~0ull : 0ull;
	if (base < n)
	{
		uint32_t p;
		v = lds_extreme<MIN, true>(A + base, n - base < CH ? n - base : CH, p, lane);
	}
	if (lane == 0) ext[c] = v;
	wave_sync();
}

// bitonic sort of the pairs (K[i], L[i]), i < P (a power of two), ascending by (K, L); one wave, arrays in HBM scratch
__device__ __forceinline__ void wide_bitonic(uint64_t *K, uint64_t *L, uint32_t P, int lane)
{
	for (uint32_t k = 2; k <= P; k <<= 1)
		for (uint32_t j = k >> 1; j > 0; j >>= 1)
		{
			for (uint32_t t = lane; t < P / 2; t += 64)
			{
				const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
				const uint32_t l = i | j;
				const uint64_t ka = ldk<true>(&K[i]), kb = ldk<true>(&K[l]);
				const uint64_t la = ldk<true>(&L[i]), lb = ldk<true>(&L[l]);
				const bool a_gt_b = ka > kb || (ka == kb && la > lb);
				const bool up = (i & k) == 0;
				if (up == a_gt_b && !(ka == kb && la == lb))
				{
					stk<true>(&K[i], kb); stk<true>(&K[l], ka);
					stk<true>(&L[i], lb); stk<true>(&L[l], la);
				}
			}
			set_sync<true>();
		}
}

template <int FUNC, typename SH>
__global__ __launch_bounds__(256) void hnsw_search_kernel_wide(const SearchArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	unsigned char *my = smem + (size_t) wib * a.wave_bytes;
	float        *qf      = reinterpret_cast<float *>(my);
	const float4 *q4      = reinterpret_cast<const float4 *>(my);
	uint64_t     *rmax    = reinterpret_cast<uint64_t *>(my + a.off_res);      // wide_nr chunk maxima of res
	uint64_t     *cmin    = reinterpret_cast<uint64_t *>(my + a.off_cand);     // wide_nc chunk minima of cand
	uint32_t     *newid   = reinterpret_cast<uint32_t *>(my + a.off_newid);
	float        *newdist = reinterpret_cast<float *>(my + a.off_newdist);

	const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wib;
	uint64_t *res  = a.set_scratch + (size_t) slot * a.set_stride;             // wide_p keys
	uint64_t *cand = res + a.wide_p;                                           // 2 * wide_p keys (capacity used: ccap)
	uint32_t *vis  = a.vis + (size_t) slot * a.vis_words;
	uint32_t *vlog = a.vlog + (size_t) slot * a.logcap;
	const uint32_t ef = a.ef, CH = a.wide_ch;
	bool aborted = false;

	for (;;)
	{
		uint32_t qi = 0;
		if (lane == 0) qi = atomicAdd(a.ticket, 1u);
		qi = __builtin_amdgcn_readfirstlane(qi);
		if (qi >= a.nq) break;
		// (an abort request is sticky for this wave: it takes the remaining tickets without walking and marks every query it does not
		// answer with count 0xFFFFFFFF, so that the caller of an interrupted launch can tell which rows of its outputs are results)
		if (!aborted && (qi & a.abort_mask) == 0u && abort_requested(a)) aborted = true;
		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }   // (wave-uniform by construction; said explicitly)
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi] = __builtin_amdgcn_s_memrealtime();

		const float *qsrc = a.queries + (size_t) qi * a.q_stride;
		for (uint32_t e = lane; e < a.qpad_floats; e += 64)
		{
			const float t = qsrc[e < a.dim ? e : a.dim - 1];
			qf[e] = (e < a.dim) ? t : 0.f;
		}
		for (uint32_t i = lane; i < a.wide_nr; i += 64) rmax[i] = 0ull;
		for (uint32_t i = lane; i < a.wide_nc; i += 64) cmin[i] = ~0ull;
		wave_sync();
		float qnorm = 0.f;
		if (FUNC == F_COSINE) qnorm = query_norm(q4, a.nchunks, a.kiters, lane);

		uint32_t rsize = 0, csize = 0, logn = 0, evals = 0, hops = 0;
		uint64_t lbkey = 0;                                   // largest result key (its distance word = lowerBound, hnswalg.cpp:65,107)

		if (a.n > 0)
		{
			const uint32_t ep = a.entry;                                   // hnswalg.cpp:55-65
			{
				auto one = [ep](uint32_t) { return ep; };
				score_rows<FUNC, SH::KB, 1>(a.vec, a.stride, q4, a.nchunks, a.kiters, one, 1u, newdist, lane);
			}
			wave_sync();
			const float d0 = finish_dist<FUNC>(newdist[0], newdist[OUT2], qnorm);
			evals = 1;
			if (a.out_evals && a.evals_cap && lane == 0) a.out_evals[(size_t) qi * a.evals_cap] = ep;
			const uint64_t hi0 = (uint64_t) ord_f32(d0) << 32;
			if (lane == 0)
			{
				stk<true>(&res[0], hi0 | ep);
				stk<true>(&cand[0], hi0 | (uint32_t) ~ep);
				rmax[0] = hi0 | ep;
				cmin[0] = hi0 | (uint32_t) ~ep;
				vis[ep >> 5] = 1u << (ep & 31);
				vlog[0] = ep;
			}
			lbkey = hi0 | ep;
			rsize = csize = logn = 1;
			set_sync<true>();

			while (csize > 0)                                               // hnswalg.cpp:67-112
			{
				// candidateSet.top(): smallest chunk minimum, then its position inside that chunk
				uint32_t cidx;
				const uint64_t ck = lds_extreme<true, false>(cmin, (csize + CH - 1) / CH, cidx, lane);
				if ((uint32_t) (ck >> 32) > (uint32_t) (lbkey >> 32)) break;   // :70-71 (ord() keeps the order of the distances)
				const uint32_t cur = ~(uint32_t) ck;
				{
					const uint32_t base = cidx * CH;
					uint32_t p;
					(void) lds_extreme<true, true>(cand + base, csize - base < CH ? csize - base : CH, p, lane);
					const uint32_t pos = base + p, last = csize - 1;
					csize = last;                                           // :73 pop = last entry into the hole
					if (pos != last && lane == 0) stk<true>(&cand[pos], ldk<true>(&cand[last]));
					set_sync<true>();
					wide_refresh<true>(cand, csize, cidx, CH, cmin, lane);
					if (last / CH != cidx) wide_refresh<true>(cand, csize, last / CH, CH, cmin, lane);
				}
				if (a.out_pops && hops < a.pops_cap && lane == 0)
					__hip_atomic_store(a.out_pops + (size_t) qi * a.pops_cap + hops, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				hops++;
				if ((hops & 255u) == 0u && abort_requested(a)) { aborted = true; break; }

				for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)               // :76-77
				{
					const uint32_t j = j0 + lane;
					const uint32_t t = a.links[(size_t) cur * a.lstride + (j < a.lstride ? j : a.lstride - 1)];
					bool isnew = false;
					if (j < a.lstride && t != LINK_NONE)                    // :91-93
					{
						const uint32_t bit = 1u << (t & 31);
						const uint32_t old = atomicOr(&vis[t >> 5], bit);
						isnew = !(old & bit);
					}
					const uint64_t mask = __ballot(isnew);
					const uint32_t nnew = (uint32_t) __builtin_popcountll(mask);
					if (nnew == 0) continue;
					const uint32_t rank = lane_rank(mask);
					if (isnew)
					{
						newid[rank] = t;
						if (a.out_evals && evals + rank < a.evals_cap) a.out_evals[(size_t) qi * a.evals_cap + evals + rank] = t;
						const uint32_t lp = logn + rank;
						if (lp < a.logcap) vlog[lp] = t;
					}
					logn += nnew;
					wave_sync();
					{
						const uint32_t *ids = newid;
						auto by_id = [ids](uint32_t r) { return ids[r]; };
						score_rows_fit<FUNC, SH::KB, SH::RPG>(a.vec, a.stride, q4, a.nchunks, a.kiters, by_id, nnew, newdist, lane);
					}
					evals += nnew;
					wave_sync();
					const uint32_t od_mine = ord_f32(finish_dist<FUNC>(newdist[lane], newdist[OUT2 + lane], qnorm));
					const uint32_t t_mine = newid[lane];
					uint64_t todo = __ballot((uint32_t) lane < nnew && (rsize < ef || (uint32_t) (lbkey >> 32) > od_mine));
					while (todo)                                            // :99-108, in link order
					{
						const uint32_t r = (uint32_t) __builtin_ctzll(todo);
						todo &= todo - 1;
						const uint32_t od = (uint32_t) __builtin_amdgcn_readlane((int) od_mine, (int) r);
						if (!(rsize < ef || (uint32_t) (lbkey >> 32) > od)) continue;
						const uint32_t t2 = (uint32_t) __builtin_amdgcn_readlane((int) t_mine, (int) r);
						const uint64_t hi = (uint64_t) od << 32;
						const uint64_t ckey = hi | (uint32_t) ~t2, rkey = hi | t2;
						if (csize == a.ccap)                                // :100; make room: drop what can never be expanded
						{
							// (only reachable with a full result set: every accepted element adds one candidate and ccap = 2 * ef)
							const uint32_t bound = (uint32_t) (lbkey >> 32);
							uint32_t w = 0;
							for (uint32_t b = 0; b < csize; b += 64)
							{
								const uint32_t i = b + lane;
								const uint64_t k = i < csize ? ldk<true>(&cand[i]) : ~0ull;
								const bool keep = i < csize && (uint32_t) (k >> 32) <= bound;
								const uint64_t km = __ballot(keep);
								set_sync<true>();                           // every lane has its key before anything at or below it is overwritten
								if (keep) stk<true>(&cand[w + lane_rank(km)], k);
								w += (uint32_t) __builtin_popcountll(km);
							}
							csize = w;
							set_sync<true>();
							for (uint32_t c = 0; c < a.wide_nc; c++) wide_refresh<true>(cand, csize, c, CH, cmin, lane);
						}
						{
							if (lane == 0)
							{
								stk<true>(&cand[csize], ckey);
								const uint32_t c = csize / CH;
								if (ckey < cmin[c]) cmin[c] = ckey;
							}
							csize++;
						}
						if (rsize < ef)                                     // :102
						{
							if (lane == 0)
							{
								stk<true>(&res[rsize], rkey);
								const uint32_t c = rsize / CH;
								if (rkey > rmax[c]) rmax[c] = rkey;
							}
							rsize++;
							lbkey = rkey > lbkey ? rkey : lbkey;
						}
						else                                                // :104-105 evict the largest, then :107
						{
							uint32_t ridx, p;
							(void) lds_extreme<false, false>(rmax, a.wide_nr, ridx, lane);
							const uint32_t base = ridx * CH;
							(void) lds_extreme<false, true>(res + base, rsize - base < CH ? rsize - base : CH, p, lane);
							if (lane == 0) stk<true>(&res[base + p], rkey);
							set_sync<true>();
							wide_refresh<false>(res, rsize, ridx, CH, rmax, lane);
							uint32_t dummy;
							lbkey = lds_extreme<false, false>(rmax, a.wide_nr, dummy, lane);
						}
						wave_sync();
					}
					set_sync<true>();
				}
			}
		}

		if (__builtin_amdgcn_readfirstlane((int) aborted)) { if (lane == 0) a.out_counts[qi] = ABORTED_COUNT; continue; }      // interrupted inside its walk
		if (a.out_times && lane == 0) a.out_times[2 * (size_t) qi + 1] = __builtin_amdgcn_s_memrealtime();
		// ---- emit: (key, label) pairs sorted by a bitonic network in the slot's scratch ----------------------------------
		const size_t obase = (size_t) qi * a.out_stride;
		uint32_t P = 1;
		while (P < rsize) P <<= 1;
		if (P < 2) P = 2;
		uint64_t *lab = cand;                                   // the candidate array is dead now (2 * wide_p keys)
		uint32_t nout = 0;
		// sort key: results that are not returned (vacuumed labels, hnsw_is_deleted, embedding.c:948-953) go behind all others
		for (uint32_t b = 0; b < P; b += 64)
		{
			const uint32_t i = b + lane;
			uint64_t k = ~0ull, l = ~0ull;
			bool keep = false;
			if (i < rsize)
			{
				k = ldk<true>(&res[i]);
				if (a.mode == 1) { l = 0; keep = true; }
				else
				{
					l = a.labels[(uint32_t) k];
					keep = !((l >> 48) & 1);
					k = keep ? (k & 0xFFFFFFFF00000000ull) : ~0ull;         // order by (dist, label), hnswalg.cpp:236,246
					l = keep ? l : ~0ull;
				}
			}
			nout += (uint32_t) __builtin_popcountll(__ballot(keep));
			if (i < P) { stk<true>(&res[i], k); stk<true>(&lab[i], l); }
		}
		set_sync<true>();
		wide_bitonic(res, lab, P, lane);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // final arrays: plain loads below
		for (uint32_t i = lane; i < a.out_stride; i += 64)
		{
			const bool in = i < nout;
			const uint64_t k = in ? ldk<true>(&res[i]) : 0;
			if (a.mode == 1) a.out_idx[obase + i] = in ? (uint32_t) k : LINK_NONE;
			else a.out_labels[obase + i] = in ? ldk<true>(&lab[i]) : ~0ull;
			if (a.out_dists) a.out_dists[obase + i] = in ? unord_f32((uint32_t) (k >> 32)) : __builtin_inff();
		}
		if (lane == 0)
		{
			a.out_counts[qi] = nout;
			if (a.out_stats) { a.out_stats[2 * (size_t) qi] = evals; a.out_stats[2 * (size_t) qi + 1] = hops; }
		}
		if (a.done) signal_done(a.done + qi, lane);

		// ---- restore the all-zero bitmap for the next query of this slot --------------
		set_sync<true>();
		if (logn <= a.logcap)
		{
			for (uint32_t i = lane; i < logn; i += 64) vis[vlog[i] >> 5] = 0u;
		}
		else
		{
			for (uint64_t w = lane; w < a.vis_words; w += 64) vis[w] = 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0);
		set_sync<true>();
	}
	if (aborted && lane == 0) atomicAdd(a.health + HEALTH_ABORTED_WAVES, 1u);
}

}  // namespace pgemb
