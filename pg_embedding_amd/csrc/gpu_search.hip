// gpu_search.hip — launch planning of the search kernels (csrc/device_search.h), the search entry points, one traced walk, search contexts
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"

// ------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------
// which kernel a launch runs: search_kernels.h (one translation unit per load shape)
static search_kernel_t pick_search_kernel(int func, uint32_t kiters, int rreg, bool team, bool narrow5, bool lean)
{
	switch (shape_index(kiters))
	{
		case 0:
			if (narrow5 && !team) return pick_kernel_shape2x2(func, rreg, lean);      // the hot narrow-row form: 8 rows per pass, 96 VGPRs, 5 waves/SIMD
			return pick_kernel_shape2x4(func, rreg, team);
		case 1:  return pick_kernel_shape4x2(func, rreg, team);
		case 2:  return pick_kernel_shape8x2(func, rreg, team);
		default:
#ifdef HNSW_EXPERIMENT
			if (knob(K_SHAPE_12X1, 0)) return pick_kernel_shape12x1(func, rreg, team);
#endif
			return pick_kernel_shape12x2(func, rreg, team);
	}
}

static const size_t VIS_BUDGET_BYTES = (size_t) 24 << 30;     // cap on bitmap workspace
static const size_t SET_BUDGET_BYTES = (size_t) 8 << 30;      // cap on the HBM result/candidate areas (generic form)
// (no cap on the effective beam: beyond WIDE_EF_MIN the wide-beam form keeps both sets with a second level of chunk extremes,
// device_search_wide.h; what bounds a beam is the per-slot scratch, 24 bytes per result slot, under SET_BUDGET_BYTES)
static const size_t WIDE_EF_MIN = 2048;

int launch_search(hnsw_gpu_index *ix, SearchWs *w, const float *d_queries, size_t q_stride, size_t nq, size_t ef, int mode,
						 uint64_t *d_labels, uint32_t *d_idx, float *d_dists, uint32_t *d_counts,
						 uint32_t *d_stats, hipStream_t stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_queries || !d_counts || (mode == 0 && !d_labels) || (mode == 1 && !d_idx))
		return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	if (nq >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "too many queries");
	// A beam wider than the index behaves exactly like a beam of the index size (nothing is ever evicted,
	// the walk ends when the candidates run out), so the scan's efSearch doubling (embedding.c:334) can go
	// as far as it likes; the output arrays keep the caller's ef as their row stride.
	const size_t out_stride = ef;
	ef = std::min(ef, std::max<size_t>(ix->n, 1));
	HIPCHK(hipSetDevice(ix->device));

	SearchArgs a;
	memset(&a, 0, sizeof(a));
	a.vec = ix->vec; a.links = ix->links; a.labels = ix->labels;
	a.n = (uint32_t) ix->n; a.dim = (uint32_t) ix->meta.dim; a.stride = ix->stride;
	a.nchunks = ix->stride / 4; a.kiters = (a.nchunks + 15) / 16;
	a.maxM = (uint32_t) ix->meta.maxM; a.lstride = ix->lstride; a.entry = ix->meta.enterpoint_node;
	a.queries = d_queries; a.q_stride = (uint32_t) q_stride; a.nq = (uint32_t) nq; a.ef = (uint32_t) ef; a.ccap = (uint32_t) (2 * ef);
	a.out_stride = (uint32_t) out_stride;
	a.out_labels = d_labels; a.out_idx = d_idx; a.out_dists = d_dists; a.out_counts = d_counts; a.out_stats = d_stats;
	a.mode = mode;
	if (a.n > 0 && a.entry >= a.n) return fail(HNSW_GPU_ERR_ARG, "enterpoint_node %u >= count %u", a.entry, a.n);

	// LDS carve per wave
	a.qpad_floats = (uint32_t) round_up(a.kiters, shape_kb(shape_index(a.kiters))) * 64;
	// Form of the accepted-set bookkeeping (rreg):
	//   beam form (one accepted set in registers, acceptance by counting): default up to ef = 256, and up
	//     to ef = 512 for rows wider than 256 floats — those run at 2 waves/SIMD anyway, so 16 set registers
	//     beat the LDS form there (+22-28 %, profiles/r1i_beam_form.txt); narrow rows keep the LDS form
	//     above 256 (it holds 4 waves/SIMD).  Its prune packs the "expanded" bit into bit 31 of the idx.
	//   LDS (generic) form: everything else — also HNSW_GPU_BEAM=0 and mirrors of >= 2^31 elements (HNSW_GPU_FORCE_LDS_HEAPS=1 forces it).
	//   (two-set register form, round 1's hot kernel: experiment builds only since round 5.)
	// The register forms use their LDS "res"/"cand" areas only as scratch of the emit step.
	knobs_init();
	const bool use_beam = knob(K_BEAM, 1) != 0 && ix->cap < 0x80000000ull;
	const bool beam16 = use_beam && ef > 256 && ef <= 512 && (knob_is_set(K_BEAM16) ? knob(K_BEAM16, 0) > 0 : shape_index(a.kiters) >= 2);
	const size_t wide_min = (size_t) knob(K_WIDE_EF_MIN, (long long) WIDE_EF_MIN);
	// Reference-order arithmetic (device_dist.h, F_L2_REF / F_MANHATTAN_REF / F_COSINE_REF; opt-in, HNSW_GPU_REF_ORDER=1): the summation order of
	// oracle/_ref's own build, so that the id lists ARE the compiled reference's, query by query.  One kernel set only: beam form, 4 set
	// registers, one wave per query; since round 6 with the canonical code's load shape (a timed mode of bench.py).
	int func_code = (int) ix->meta.dist_func;
	bool reforder = false;
	if (knob(K_REF_ORDER, 0) > 0)
		{
			const bool ok = ef <= 128 && ix->cap < 0x80000000ull &&
							((func_code == F_L2 && ix->meta.dim % 16 == 0) || ((func_code == F_MANHATTAN || func_code == F_COSINE) && ix->meta.dim % 4 == 0));
			if (!ok)
				return fail(HNSW_GPU_ERR_ARG, "HNSW_GPU_REF_ORDER: only L2 with dims %% 16 == 0 or cosine / Manhattan with dims %% 4 == 0, ef <= 128");
			reforder = true;
			func_code = func_code == F_L2 ? F_L2_REF : (func_code == F_COSINE ? F_COSINE_REF : F_MANHATTAN_REF);
		}
	int rreg;
	if (reforder) rreg = -4;
	else if (ef > wide_min) rreg = 3;
	else if (knob(K_FORCE_LDS_HEAPS, 0) > 0) rreg = 0;
	else if (use_beam && (ef <= 256 || beam16)) rreg = ef <= 64 ? -2 : (ef <= 128 ? -4 : (ef <= 256 ? -8 : -16));
#ifdef HNSW_EXPERIMENT
	else rreg = ef <= 128 ? 2 : (ef <= 256 ? 4 : 0);         // two-set register form (experiment builds)
#else
	else rreg = 0;                                           // HNSW_GPU_BEAM=0, or a mirror of >= 2^31 elements: the generic form
#endif
	const size_t ucap = rreg < 0 ? (size_t) 64 * (size_t) -rreg : 0;      // beam form: slots of the accepted set
	// Team form wanted for this launch?  (decided for good further down, once the LDS carve is known)
	const int treq = (int) knob(K_TEAM, -1);
	const size_t auto_nq = (size_t) knob(K_TEAM_MAX_NQ, (long long) ix->num_cu);
	const bool stream_launch = w->stream_host_next != nullptr;
	const bool team_wanted = rreg < 0 && !reforder && (stream_launch || (treq != 0 && (treq > 0 || ix->stride > 320 || nq <= auto_nq)));
	// narrow rows, hot form: beam kernel with <= 4 set registers, one sum per row (L2 / Manhattan), not a team
	const bool narrow5 = shape_index(a.kiters) == 0 && (rreg == -2 || rreg == -4) && (int) ix->meta.dist_func != F_COSINE &&
						 !team_wanted && !reforder && knob(K_NARROW5, 1) != 0;
	size_t off = (size_t) a.qpad_floats * 4;
	// reference-order arithmetic: the per-wave stage of its transposed accumulation sits right behind the query image (device_dist.h, score_rows_ref)
	if (reforder) off += (size_t) 8 * (func_code == F_COSINE_REF ? 2 : 1) * REF_STAGE_ROW * 4;
	if (rreg == 3)
	{
		// wide-beam form: both sets in the slot's HBM area [res: P | cand: 2P], P = the power of two >= ef (the output sort is a
		// bitonic network); per-chunk extremes in LDS: chunks of >= 1024 keys, at most 1024 chunks of candidates
		size_t P = 2;
		while (P < ef) P <<= 1;
		size_t ch = 1024;
		while ((2 * ef + ch - 1) / ch > 1024) ch <<= 1;
		a.wide_p = (uint32_t) P; a.wide_ch = (uint32_t) ch;
		a.wide_nr = (uint32_t) ((ef + ch - 1) / ch); a.wide_nc = (uint32_t) ((2 * ef + ch - 1) / ch);
		a.set_stride = 3 * P;
		a.off_res = (uint32_t) off;  off += round_up((size_t) a.wide_nr * 8, 16);
		a.off_cand = (uint32_t) off; off += round_up((size_t) a.wide_nc * 8, 16);
		if (3 * P * 8 > SET_BUDGET_BYTES)
			return fail(HNSW_GPU_ERR_NOMEM, "ef %zu needs %zu bytes of scratch per query slot (more than the %zu-byte budget)", ef, 3 * P * 8, SET_BUDGET_BYTES);
	}
	else if (rreg)
	{
		// [query | hash set (overlaid by the emit step's tie scratch) | newid | newdist]
		const size_t fixed = off + 64 * 4 + 128 * 4 + (team_wanted ? sizeof(TeamCtl) : 0);   // (+ the wave's control block behind the regions)
		// Rows of >= 1.25 KiB make the traversal HBM-bound, and there the LDS set pays (no L2
		// atomics, ~10 % less HBM traffic; measured 5.4 -> 7.5 TB/s at 768 dims) at 8 waves per CU.
		// Narrow rows are latency-bound and want 16-20 waves per CU: the beam form gives them a bucketed set of
		// 3456-4096 16-bit tags (ids whose bucket is full go to the bitmap); the two-set register form keeps the
		// bitmap only (profiles/r1g_visited_set_by_dim.txt, profiles/r1i_beam_form.md, profiles/r2m_*).
		// (rows of up to 128 floats in the beam form with ef <= 128, L2 / Manhattan, launches that will not run as
		// teams: 5 waves/SIMD with the 8-rows-per-pass shape — measured +6-10 % over 4 waves, profiles/r2m_*)
		const bool wide = ix->stride > 320;
		size_t want_waves = wide ? 8 : (narrow5 ? 20 : 16);
#ifdef HNSW_EXPERIMENT
		if (wide && knob(K_WIDE_WAVES, 0) >= 4) want_waves = (size_t) knob(K_WIDE_WAVES, 0);   // (experiment builds at 3 waves/SIMD)
#endif
		uint32_t hcap = wide ? 4096 : (rreg < 0 ? 2048 : 0);
		if (knob_is_set(K_HASH_ENTRIES)) hcap = (uint32_t) knob(K_HASH_ENTRIES, 0);
		// emit scratch: [keys | labels]; the beam form sorts up to `ucap` survivors (ties at the bound)
		const size_t nkeys = ucap ? ucap : ef;
		const size_t emit = round_up(nkeys * 8, 16) + round_up(ef * 8, 16);
		if (rreg < 0)
		{
			// beam form: hcap/4 buckets (any count, 128-byte steps of LDS) of eight 16-bit tags (device_search.h,
			// "bucketed"); tag = id / buckets + 1 must fit 16 bits and the 38-bit reciprocal must be exact (ids below
			// 2^28), else the kernel runs on the HBM bitmap alone
			hcap = std::min<uint32_t>(hcap, 4096);
			while (hcap >= 512 && want_waves * (fixed + std::max<size_t>(hcap * 4, emit)) > LDS_PER_CU) hcap -= 128;
			hcap &= ~31u;
			if (hcap < 512 || (uint64_t) ix->cap > (uint64_t) 65535 * (hcap / 4) || ix->cap >= (1u << 28)) hcap = 0;
			a.hmagic = hcap ? (uint32_t) ((((uint64_t) 1 << 38) + hcap / 4 - 1) / (hcap / 4)) : 0;
		}
		else
		{
			while (hcap >= 512 && want_waves * (fixed + std::max<size_t>(hcap * 4, emit)) > LDS_PER_CU) hcap >>= 1;
			if (hcap < 512 || (hcap & (hcap - 1))) hcap = 0;
		}
		a.hcap = hcap;
		a.hmax = hcap - hcap / 4;
		a.off_hash = (uint32_t) off;
		a.off_res = (uint32_t) off;
		a.off_cand = (uint32_t) (off + round_up(nkeys * 8, 16));
		off += round_up(std::max<size_t>((size_t) hcap * 4, emit), 16);
	}
	else
	{
		// generic form: [res ef+1 | cand 2ef+1] keys per wave — in LDS while at least HNSW_GPU_LDS_SET_MIN_WAVES
		// (default 4) waves per CU fit, otherwise in a per-slot HBM area (any ef)
		const size_t set_bytes = round_up((ef + 1) * 8, 16) + round_up((2 * ef + 1) * 8, 16);
		const size_t min_waves = knob(K_LDS_SET_MIN_WAVES, 0) > 0 ? (size_t) knob(K_LDS_SET_MIN_WAVES, 0) : 4;
		if (min_waves * (off + set_bytes + 64 * 4 + 128 * 4) > LDS_PER_CU)
		{
			rreg = 1;
			a.off_res = 0;
			a.off_cand = (uint32_t) (ef + 1);                     // in keys, inside the slot's area
			a.set_stride = 3 * ef + 2;
		}
		else
		{
			a.off_res = (uint32_t) off;     off += round_up((ef + 1) * 8, 16);
			a.off_cand = (uint32_t) off;    off += round_up((2 * ef + 1) * 8, 16);
		}
	}
	a.off_newid = (uint32_t) off;   off += 64 * 4;
	a.off_newdist = (uint32_t) off; off += 128 * 4;      // sums + (cosine) |x|^2
	a.wave_bytes = (uint32_t) round_up(off, 16);
	if (a.wave_bytes > LDS_PER_CU)
		return fail(HNSW_GPU_ERR_ARG, "ef=%zu dim=%zu needs %u bytes of LDS per query (> %zu)", ef, ix->meta.dim,
					a.wave_bytes, LDS_PER_CU);
	uint32_t wpb = 4;
	while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1;
	// Team form of the beam kernel (device_search.h, "Team form"): waves of a block that have no query (left) help a
	// sibling's walk with packages prepared in their own, otherwise idle LDS regions.  Measured at 1M rows
	// (profiles/r2_team_form.txt): rows wider than 320 floats gain at every launch size (one query 0.68 -> 0.47 ms,
	// 256 queries -24 %, 10 000 -3 %, 40 000 -0.7 %: only the tail of a big launch has idle waves); narrow rows gain
	// up to ~256 queries per launch and lose beyond (the larger kernel costs the 4-waves-per-SIMD steady state 10-16 %).
	// HNSW_GPU_TEAM=0/1 forces it off/on, HNSW_GPU_TEAM_MAX_NQ moves the narrow-row threshold, HNSW_GPU_TEAM_WPB the
	// waves per block (default 8 when the LDS of a block allows).
	bool team = false;
	if (rreg < 0)
	{
		const size_t pub = (size_t) 64 * ucap / 64 * 8;                 // 64*UREG keys
		// a donated region: [accepted-set copy | expanded bits | miss ids | package headers | packages | memo], all below
		// off_newid.  As many package slots as leave a useful memo: an element packaged while it was 6th in line may
		// be popped dozens of hops later, and a direct-mapped slot that was reused by then is a lost package.
		uint32_t lcs = 32;
		size_t o_ex = pub, o_miss = o_ex + 256, o_tag = round_up(o_miss + 256, 8), o_state = 0, o_links = 0, o_dc = 0, dccap = 0;
		for (; lcs >= 4; lcs >>= 1)
		{
			o_state = o_tag; o_links = o_tag + lcs * 8;                          // headers: lcs x u64; packages: lcs x lstride x u64
			o_dc = round_up(o_links + (size_t) lcs * a.lstride * 8, 16);
			const size_t want = lcs >= 16 ? 512 : (lcs == 8 ? 256 : 128);         // memo entries this many slots must leave
			dccap = 0;
			if (o_dc + want * 8 <= a.off_newid)
			{
				dccap = want;
				while (o_dc + dccap * 2 * 8 <= a.off_newid && dccap < 2048) dccap *= 2;
				break;
			}
		}
		if (dccap >= 128 && team_wanted)
		{
			team = true;
			a.tm_off_ex = (uint32_t) o_ex; a.tm_off_miss = (uint32_t) o_miss; a.tm_off_lctag = (uint32_t) o_tag;
			a.tm_off_lcstate = (uint32_t) o_state; a.tm_off_lclinks = (uint32_t) o_links; a.tm_lcslots = lcs;
			a.tm_off_dc = (uint32_t) o_dc; a.tm_dccap = (uint32_t) dccap;
			{
				// helpers of rank < tm_spec prepare packages ahead of the walk; the others score slices of its many-row hops
				// (device_search.h, banner at TeamCtl).  Measured at 1M rows (profiles/r3a_slice_helpers.txt): 768 dims, 5 of
				// 7 helpers speculating: one query 0.470 -> 0.438 ms, 16 queries -3.4 %, 256 -4.1 %, 1024 -3.4 %, 10 000 -0.5 %,
				// 40 000 -0.2 %; 3: 0.452; 0 (nobody speculates): 0.618.  128 dims: a hop rarely has more rows than one pass of 16,
				// slices lose 1-2 %, so narrow rows let every helper speculate.  HNSW_GPU_TEAM_SPEC overrides (8 = all speculate).
				a.tm_spec = knob_is_set(K_TEAM_SPEC) ? (uint32_t) std::max<long long>(0, knob(K_TEAM_SPEC, 0)) : (ix->stride > 320 ? 5u : 8u);
			}
			int maxlds = 64 * 1024;
			(void) hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, ix->device);
			uint32_t want = knob(K_TEAM_WPB, 0) > 0 ? (uint32_t) knob(K_TEAM_WPB, 0) : 8u;
			want = std::min(want, 8u);
			wpb = std::max<uint32_t>(1, (uint32_t) std::min<size_t>(want, ((size_t) maxlds - 8 * sizeof(TeamCtl)) / a.wave_bytes));
			if (wpb < 2) team = false;
		}
	}
	if (!team) { wpb = 4; while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1; }
	// (test knob: waves per block of the narrow-row one-wave launches.  A block's LDS and wave slots are released when its LAST wave
	// ends, so with 4-wave blocks a draining launch holds resources a following launch could use — profiles/r4c_*; 1 = every wave is a
	// block of its own.  Measured round 6, profiles/r6d_*.)
	if (!team && narrow5 && knob(K_NARROW_WPB, 0) > 0) wpb = (uint32_t) std::min<long long>(4, knob(K_NARROW_WPB, 0));
	a.off_ctl = (uint32_t) ((size_t) wpb * a.wave_bytes);
	const size_t lds = (size_t) wpb * a.wave_bytes + (team ? wpb * sizeof(TeamCtl) : 0);
	const bool lean = narrow5 && !team && !w->pops_next && !w->evals_next && !w->times_next && knob(K_LEAN, 1) != 0;
	search_kernel_t kern = pick_search_kernel(func_code, a.kiters, rreg, team, narrow5, lean);
	if (!kern) return fail(HNSW_GPU_ERR_INTERNAL, "no kernel for this configuration");
	{
		static const char *const shapes[4] = { "Shape2x4", "Shape4x2", "Shape8x2", "Shape12x2" };
		const char *shp = shapes[shape_index(a.kiters)];
#ifdef HNSW_EXPERIMENT
		if (shape_index(a.kiters) == 3 && knob(K_SHAPE_12X1, 0)) shp = "Shape12x1";
#endif
		if (narrow5 && !team) shp = "Shape2x2";
		if (rreg < 0) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_beam<%d, pgemb::%s, %d, %s, %s>", func_code, shp, -rreg, team ? "true" : "false", lean ? "true" : "false");
		else if (rreg == 3) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_wide<%d, pgemb::%s>", (int) ix->meta.dist_func, shp);
		else if (rreg >= 2) snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_reg<%d, pgemb::%s, %d>", (int) ix->meta.dist_func, shp, rreg);
		else snprintf(w->kname, sizeof(w->kname), "pgemb::hnsw_search_kernel_lds<%d, pgemb::%s, %s>", (int) ix->meta.dist_func, shp, rreg == 1 ? "true" : "false");
	}
	if (lds > 48 * 1024)
		HIPCHK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	int per_cu = 0;
	HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int) (wpb * 64), lds));
	if (per_cu < 1) per_cu = 1;
	if (knob(K_BLOCKS_PER_CU, 0) > 0) per_cu = std::min(per_cu, (int) knob(K_BLOCKS_PER_CU, 0));
	size_t blocks = std::min<size_t>((nq + wpb - 1) / wpb, (size_t) per_cu * ix->num_cu);
	a.team_mains = wpb;
	if (team && nq < (size_t) per_cu * ix->num_cu * wpb)
	{
		// fewer queries than resident waves: spread them over the blocks, the other waves of a block start as helpers
		blocks = std::min<size_t>(nq, (size_t) per_cu * ix->num_cu);
		a.team_mains = (uint32_t) std::min<size_t>(wpb, (nq + blocks - 1) / blocks);
		// ... unless the caller knows better: a host that keeps SEVERAL small launches in flight (the batching server's lanes) says
		// how many waves of a block should walk — one walk per 8-wave block is the latency shape of a lone launch; six such launches
		// of 190 queries want 9 000 waves of a device that holds 2 048, i.e. at most 256 walks run at a time however many wait
		// (profiles/r4d_server_sweep.txt: the server's 0.54 M q/s ceiling is exactly 256 walks of 0.47 ms)
		if (w->walkers_hint > a.team_mains)
		{
			a.team_mains = std::min<uint32_t>(wpb, w->walkers_hint);
			blocks = std::min<size_t>((nq + a.team_mains - 1) / a.team_mains, (size_t) per_cu * ix->num_cu);
		}
	}
	// (experiment knob: walking waves per block of a team launch — the others help from the start; scripts/exp_spec_ab.py)
#ifdef HNSW_EXPERIMENT
	if (team && knob(K_TEAM_MAINS, 0) > 0) a.team_mains = std::min<uint32_t>(a.team_mains, (uint32_t) knob(K_TEAM_MAINS, 0));
#endif
	if (stream_launch)
	{
		// a resident launch fed by the host (device_search.h, "Stream mode"): exactly the blocks the device holds at once — block 0 is the
		// doorbell and must be resident for any other block to make progress
		if (!team) { w->stream_host_next = nullptr; return fail(HNSW_GPU_ERR_ARG, "a stream needs the team form of the beam kernel (ef <= 256, or <= 512 on wide rows)"); }
		blocks = std::max<size_t>(2, (size_t) per_cu * ix->num_cu);
		a.team_mains = std::min<uint32_t>(wpb, std::max<uint32_t>(1u, w->stream_walkers_next));
		a.stream_host = w->stream_host_next; a.stream_dev = w->stream_dev_next; a.stream_ring = w->stream_ring_next;
		a.stream_light = knob(K_STREAM_LIGHT, 1) != 0 ? 1u : 0u;
		w->stream_host_next = nullptr; w->stream_dev_next = nullptr;
	}
	// (test knob: fewer blocks than the launch would get, so that the waves with queries take SEVERAL each through the
	// ticket counter while their siblings help — the schedule of a small launch whose other blocks start late,
	// tests/experiments/team_second_walk_stress.py)
	if (knob(K_MAX_BLOCKS, 0) > 0) blocks = std::min<size_t>(blocks, (size_t) knob(K_MAX_BLOCKS, 0));

	// workspace: one bitmap + log per resident wave
	const size_t words = std::max<size_t>(1, (ix->cap + 31) / 32);   // by capacity: stable while the index grows
	size_t max_slots = std::max<size_t>(wpb, VIS_BUDGET_BYTES / (words * 4));
	if (rreg == 1 || rreg == 3) max_slots = std::max<size_t>(wpb, std::min(max_slots, SET_BUDGET_BYTES / (a.set_stride * 8)));
	if (blocks * wpb > max_slots) blocks = std::max<size_t>(1, max_slots / wpb);
	const size_t slots = blocks * wpb;
	const uint32_t logcap = 8192;
	if (slots > w->vis_slots || words != w->vis_words)
	{
		if (w->vis) (void) hipFree(w->vis);
		if (w->vlog) (void) hipFree(w->vlog);
		w->vis = nullptr; w->vlog = nullptr; w->vis_slots = 0;
		HIPCHK(hipMalloc(&w->vis, slots * words * 4));
		HIPCHK(hipMalloc(&w->vlog, slots * (size_t) logcap * 4));
		HIPCHK(hipMemsetAsync(w->vis, 0, slots * words * 4, stream));
		w->vis_slots = slots; w->vis_words = words; w->logcap = logcap;
	}
	if (__atomic_load_n(&w->abort_sent, __ATOMIC_SEQ_CST))
	{
		// the previous launch of this workspace was asked to end early: its waves left their bitmaps as they were
		fprintf(stderr, "pg_embedding_amd: the previous search launch of this workspace (%s) was asked to end early (abort word): the queries it did "
				"not answer have count HNSW_GPU_COUNT_ABORTED; the workspace is re-zeroed\n", w->kname);
		HIPCHK(hipStreamSynchronize(stream));
		if (stream) HIPCHK(hipStreamSynchronize(nullptr));
		if (w->vis) HIPCHK(hipMemset(w->vis, 0, w->vis_slots * w->vis_words * 4));
		__atomic_store_n(w->abort_host, 0u, __ATOMIC_SEQ_CST);
		__atomic_store_n(&w->abort_sent, 0, __ATOMIC_SEQ_CST);
	}
	a.health = w->health; a.abort_word = w->abort_host;
	// a wave reads the abort word (pinned host memory: a read across the host link) at the top of every 2^k-th query, k = 4 (test knob:
	// 0 = every query, rounds 1-5 — the read rate of a narrow-row launch then depends on where the host serves that page from, banner at
	// abort_requested in device_search.h)
	a.abort_mask = (1u << (uint32_t) std::min<long long>(16, std::max<long long>(0, knob(K_ABORT_POLL_LOG2, 4)))) - 1u;
	a.vis = w->vis; a.vis_words = words; a.vlog = w->vlog; a.logcap = w->logcap;
	// (the beam form's prune scratch: read only by -DHNSW_OLD_COMPACT builds since round 6 — the shipped kernels compact in registers; 2 KB per slot)
	if (ucap && slots * ucap > w->beam_keys)
	{
		if (w->beam) (void) hipFree(w->beam);
		w->beam = nullptr; w->beam_keys = 0;
		HIPCHK(hipMalloc(&w->beam, slots * ucap * 8));
		w->beam_keys = slots * ucap;
	}
	a.beam_scratch = w->beam;
	if (rreg == 1 || rreg == 3)
	{
		const size_t keys = slots * a.set_stride;
		if (keys > w->set_keys)
		{
			if (w->sets) (void) hipFree(w->sets);
			w->sets = nullptr; w->set_keys = 0;
			HIPCHK(hipMalloc(&w->sets, keys * 8));
			w->set_keys = keys;
		}
		a.set_scratch = w->sets;
	}
	a.ticket = w->ticket;
#ifdef HNSW_EXPERIMENT
	if (knob(K_TEAM_COUNTERS, 0))                           // (diagnostic builds only: build.py variant ... HNSW_HOP_STAMPS / HNSW_TEAM_COUNTERS)
	{
		if (!w->team_dbg) HIPCHK(hipMalloc(&w->team_dbg, 64));
		HIPCHK(hipMemsetAsync(w->team_dbg, 0, 64, stream));
		a.team_dbg = w->team_dbg;
	}
#endif
	a.done = w->done_next;
	w->done_next = nullptr;
	a.out_pops = w->pops_next; a.pops_cap = w->pops_cap_next;
	w->pops_next = nullptr; w->pops_cap_next = 0;
	a.out_evals = w->evals_next; a.evals_cap = w->evals_cap_next; a.out_times = w->times_next;
	w->evals_next = nullptr; w->evals_cap_next = 0; w->times_next = nullptr;
	HIPCHK(hipMemsetAsync(w->ticket, 0, 8, stream));

	const int evi = (int) (w->launches % SearchWs::EV_RING);
	HIPCHK(hipEventRecord(w->ev0[evi], stream));
	// (a stream is resident by design: the library's watchdog does not time it — its host stops it, hnsw_gpu_stream_close)
	__atomic_store_n(&w->busy_since_ms, stream_launch ? (int64_t) 0 : now_ms(), __ATOMIC_SEQ_CST);
	hipLaunchKernelGGL(kern, dim3((uint32_t) blocks), dim3(wpb * 64), lds, stream, a);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(w->ev1[evi], stream));
	__atomic_store_n(&w->launches, w->launches + 1, __ATOMIC_SEQ_CST);
	w->last_slots = (uint32_t) slots;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_batch_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										 label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										 void *stream)
{
	return launch_search(ix, ix ? &ix->ws : nullptr, d_queries, ix ? ix->meta.dim : 0, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats, (hipStream_t) stream);
}

extern "C" int hnsw_gpu_search_base_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t ef,
										idx_t *d_idx, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										void *stream)
{
	return launch_search(ix, ix ? &ix->ws : nullptr, d_queries, ix ? ix->meta.dim : 0, nq, ef, 1, nullptr, d_idx, d_dists, d_counts, d_stats, (hipStream_t) stream);
}

// Poll a completion flag the kernel stores into pinned host memory.  0 = set; otherwise an error: the kernel ended
// without storing it, or it has not stored it within two minutes (HNSW_GPU_POLL_LIMIT_S; a walk is under a
// millisecond: the device is hung, and polling for ever would hang the caller with it).
int poll_limit_s()
{
	knobs_init();
	return knob(K_POLL_LIMIT_S, 0) > 0 ? (int) knob(K_POLL_LIMIT_S, 0) : 120;
}

// `w` = the search workspace whose launch is waited for, or nullptr when the wait is for kernels that do not read an abort word
// (the insert kernels): on a time-out only THAT workspace is asked to end — other mirrors, contexts and shards of the process keep
// their launches (an abort makes a launch's outputs undefined).
int poll_done_flag(const volatile uint32_t *flag, const char *what, SearchWs *w)
{
	uint64_t spins = 0;
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while (*flag == 0)
	{
		__builtin_ia32_pause();
		if ((++spins & 0xFFFF) == 0)
		{
			if (hipStreamQuery(nullptr) != hipErrorNotReady)
			{
				HIPCHK(hipStreamSynchronize(nullptr));                  // the kernel is gone: either it has just stored the flag, or it died
				if (*flag == 0) return fail(HNSW_GPU_ERR_INTERNAL, "search kernel ended without completing %s", what);
				break;
			}
			struct timespec t1;
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (t1.tv_sec - t0.tv_sec > poll_limit_s())
			{
				// ask THIS launch to end (every wave looks at the abort word between queries and every 256 hops), so that the
				// device is usable again even though this call fails
				if (w) { std::lock_guard<std::mutex> g(g_ws_mu); (void) abort_ws_locked(w); }
				return fail(HNSW_GPU_ERR_INTERNAL, "kernel did not complete %s within %d s%s", what, poll_limit_s(), w ? " (its search launch was asked to end)" : "");
			}
		}
	}
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_batch(hnsw_gpu_index *ix, const coord_t *queries, size_t nq, size_t ef,
									 label_t *labels, dist_t *dists, uint32_t *counts)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	HIPCHK(hipSetDevice(ix->device));
	const size_t dim = ix->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	int rc;
	// A few queries per call — the reference's own shape is ONE (hnsw_search, embedding.c:317) — are pure latency: the
	// walk is ~0.45 ms and four blocking copies plus a stream wait added ~50 us to it.  Here the kernel reads the
	// queries from pinned host memory, writes results and per-query completion flags (system-scope release, the
	// server's streamed-completion mechanism) straight back into it, and the calling core polls the flags: no copy
	// engine, no interrupt wake-up.  The launch stays on the default stream, so whatever touches this mirror next is
	// ordered behind the kernel's last instruction, not behind the flags.
	const size_t fb = round_up(nq * 4, 256);
	if (nq <= 16 && qb + lb + db + cb + fb <= ((size_t) 4 << 20) && (knobs_init(), knob(K_NO_POLL, 0) == 0))
	{
		if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes these buffers
		if (ix->pin_bytes < qb + lb + db + cb + fb)
		{
			if (ix->pin) (void) hipHostFree(ix->pin);
			ix->pin = nullptr; ix->pin_bytes = 0;
			HIPCHK(hipHostMalloc((void **) &ix->pin, qb + lb + db + cb + fb, hipHostMallocDefault));
			ix->pin_bytes = qb + lb + db + cb + fb;
		}
		char *h = ix->pin;
		float *hq = (float *) h; uint64_t *hl = (uint64_t *) (h + qb); float *hd = (float *) (h + qb + lb);
		uint32_t *hc = (uint32_t *) (h + qb + lb + db);
		volatile uint32_t *hf = (volatile uint32_t *) (h + qb + lb + db + cb);
		memcpy(hq, queries, nq * dim * 4);
		for (size_t i = 0; i < nq; i++) hf[i] = 0;
		ix->ws.done_next = (uint32_t *) hf;
		rc = launch_search(ix, &ix->ws, hq, dim, nq, ef, 0, hl, nullptr, hd, hc, nullptr, nullptr);
		ix->ws.done_next = nullptr;
		if (rc) return rc;
		for (size_t i = 0; i < nq; i++)
		{
			rc = poll_done_flag(hf + i, "a query", &ix->ws);
			if (rc) return rc;
		}
		memcpy(labels, hl, nq * ef * 8);
		if (dists) memcpy(dists, hd, nq * ef * 4);
		memcpy(counts, hc, nq * 4);
		return HNSW_GPU_OK;
	}
	rc = ensure_scratch(ix, qb + lb + db + cb);
	if (rc) return rc;
	char *p = (char *) ix->scratch;
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	if (!ix->hb0) { HIPCHK(hipEventCreate(&ix->hb0)); HIPCHK(hipEventCreate(&ix->hb1)); }
	ix->hb_valid = false;
	HIPCHK(hipEventRecord(ix->hb0, nullptr));
	HIPCHK(hipMemcpy(dq, queries, nq * dim * 4, hipMemcpyHostToDevice));
	rc = launch_search(ix, &ix->ws, dq, dim, nq, ef, 0, dl, nullptr, dd, dc, nullptr, nullptr);
	if (rc) return rc;
	HIPCHK(hipMemcpy(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost));
	if (dists) HIPCHK(hipMemcpy(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(counts, dc, nq * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipEventRecord(ix->hb1, nullptr));
	ix->hb_valid = true;
	for (size_t i = 0; i < nq; i++)
		if (counts[i] == ABORTED_COUNT)
			return fail(HNSW_GPU_ERR_INTERNAL, "the search launch was asked to end early (abort word): query %zu has no result", i);
	return HNSW_GPU_OK;
}

// One query with its walk: results as hnsw_gpu_search_batch gives them, plus the sequence of elements the walk expanded
// (hnswalg.cpp:73) and its evaluation count.  Host pointers; the polled zero-copy mechanics of the few-queries path.
// Three steps so that a caller can consume the sequence WHILE the walk runs (the kernel stores each pop with system
// scope into pinned host memory): begin = launch, poll = the pops that have become visible since the last poll,
// end = wait + results.  One trace at a time per mirror, from one thread; no library lock is held between the steps
// (the caller may run host callbacks that leave by longjmp in between: a trace that is never ended is waited for by the
// next begin).
static const uint32_t POP_NONE = 0xFFFFFFFFu;

struct TraceLayout { size_t qb, lb, db, cb, sb, pb, fb; };
static TraceLayout trace_layout(size_t dim, size_t ef, size_t pops_cap)
{
	TraceLayout t;
	t.qb = round_up(dim * 4, 256); t.lb = round_up(ef * 8, 256); t.db = round_up(ef * 4, 256); t.cb = 256; t.sb = 256;
	t.pb = round_up(pops_cap * 4, 256); t.fb = 256;
	return t;
}

extern "C" int hnsw_gpu_search_trace_begin(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, size_t pops_cap)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (!query || pops_cap == 0) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	if (pops_cap > ((size_t) 1 << 24)) return fail(HNSW_GPU_ERR_ARG, "pops_cap %zu too large", pops_cap);
	HIPCHK(hipSetDevice(ix->device));
	if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes its buffers
	const size_t dim = ix->meta.dim;
	const TraceLayout t = trace_layout(dim, ef, pops_cap);
	const size_t need = t.qb + t.lb + t.db + t.cb + t.sb + t.pb + t.fb;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, need, hipHostMallocDefault));
		ix->pin_bytes = need;
	}
	char *h = ix->pin;
	float *hq = (float *) h; uint64_t *hl = (uint64_t *) (h + t.qb); float *hd = (float *) (h + t.qb + t.lb);
	uint32_t *hc = (uint32_t *) (h + t.qb + t.lb + t.db), *hs = (uint32_t *) (h + t.qb + t.lb + t.db + t.cb),
			 *hp = (uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb);
	volatile uint32_t *hf = (volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	memcpy(hq, query, dim * 4);
	memset(hp, 0xFF, pops_cap * 4);                 // POP_NONE: a slot the walk has not reached yet
	hf[0] = 0;
	ix->ws.done_next = (uint32_t *) hf;
	ix->ws.pops_next = hp; ix->ws.pops_cap_next = (uint32_t) pops_cap;
	int rc = base ? launch_search(ix, &ix->ws, hq, dim, 1, ef, 1, nullptr, (uint32_t *) hl, hd, hc, hs, nullptr)
				  : launch_search(ix, &ix->ws, hq, dim, 1, ef, 0, hl, nullptr, hd, hc, hs, nullptr);
	ix->ws.done_next = nullptr; ix->ws.pops_next = nullptr; ix->ws.pops_cap_next = 0;
	if (rc) return rc;
	ix->trace_active = true; ix->trace_ef = ef; ix->trace_base = base; ix->trace_cap = pops_cap; ix->trace_seen = 0;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace_poll(hnsw_gpu_index *ix, idx_t *pops, size_t max, size_t *got, int *finished)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !pops || !got || !finished) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (!ix->trace_active) return fail(HNSW_GPU_ERR_ARG, "no trace in flight");
	const TraceLayout t = trace_layout(ix->meta.dim, ix->trace_ef, ix->trace_cap);
	char *h = ix->pin;
	const volatile uint32_t *hp = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb);
	const volatile uint32_t *hf = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	const bool done = hf[0] != 0;                   // read BEFORE the scan: everything the walk stored precedes the flag
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	size_t k = 0;
	while (k < max && ix->trace_seen < ix->trace_cap)
	{
		const uint32_t v = hp[ix->trace_seen];
		if (v == POP_NONE) break;
		pops[k++] = v;
		ix->trace_seen++;
	}
	*got = k;
	*finished = (done && (ix->trace_seen >= ix->trace_cap || hp[ix->trace_seen] == POP_NONE)) ? 1 : 0;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace_end(hnsw_gpu_index *ix, label_t *labels, dist_t *dists, uint32_t *count, uint32_t *npops,
										 uint32_t *nevals)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !labels || !count || !npops) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (!ix->trace_active) return fail(HNSW_GPU_ERR_ARG, "no trace in flight");
	HIPCHK(hipSetDevice(ix->device));
	const size_t ef = ix->trace_ef;
	const TraceLayout t = trace_layout(ix->meta.dim, ef, ix->trace_cap);
	char *h = ix->pin;
	const uint64_t *hl = (const uint64_t *) (h + t.qb); const float *hd = (const float *) (h + t.qb + t.lb);
	const uint32_t *hc = (const uint32_t *) (h + t.qb + t.lb + t.db), *hs = (const uint32_t *) (h + t.qb + t.lb + t.db + t.cb);
	const volatile uint32_t *hf = (const volatile uint32_t *) (h + t.qb + t.lb + t.db + t.cb + t.sb + t.pb);
	{
		const int prc = poll_done_flag(hf, "the traced query", &ix->ws);
		ix->trace_active = false;
		if (prc) return prc;
	}
	if (ix->trace_base) { const uint32_t *hi = (const uint32_t *) hl; for (size_t i = 0; i < ef; i++) labels[i] = hi[i]; }
	else memcpy(labels, hl, ef * 8);
	if (dists) memcpy(dists, hd, ef * 4);
	*count = hc[0];
	*npops = hs[1];
	if (nevals) *nevals = hs[0];
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_trace(hnsw_gpu_index *ix, const coord_t *query, size_t ef, int base, label_t *labels, dist_t *dists,
									 uint32_t *count, idx_t *pops, size_t pops_cap, uint32_t *npops, uint32_t *nevals)
{
	if (!ix || !pops || !npops) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::unique_lock<std::recursive_mutex> lock_(ix->mu);          // the three steps as one
	int rc = hnsw_gpu_search_trace_begin(ix, query, ef, base, pops_cap);
	if (rc) return rc;
	rc = hnsw_gpu_search_trace_end(ix, labels, dists, count, npops, nevals);
	if (rc) return rc;
	const TraceLayout t = trace_layout(ix->meta.dim, ef, pops_cap);
	memcpy(pops, ix->pin + t.qb + t.lb + t.db + t.cb + t.sb, std::min<size_t>(*npops, pops_cap) * 4);
	return HNSW_GPU_OK;
}

int ws_search_ms(int device, SearchWs *w, unsigned back, float *ms)
{
	if (back >= (unsigned) SearchWs::EV_RING || (uint64_t) back >= w->launches)
		return fail(HNSW_GPU_ERR_ARG, "no record of the search launch %u launches ago", back);
	HIPCHK(hipSetDevice(device));
	const int evi = (int) ((w->launches - 1 - back) % SearchWs::EV_RING);
	HIPCHK(hipEventSynchronize(w->ev1[evi]));
	HIPCHK(hipEventElapsedTime(ms, w->ev0[evi], w->ev1[evi]));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_search_ms(hnsw_gpu_index *ix, unsigned back, float *ms)
{
	if (!ix || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return ws_search_ms(ix->device, &ix->ws, back, ms);
}

extern "C" int hnsw_gpu_last_search_ms(hnsw_gpu_index *ix, float *ms) { return hnsw_gpu_search_ms(ix, 0, ms); }

// Where the time of the last hnsw_gpu_search_batch call (host pointers, copy path: more than 16 queries) went on the device:
// out[0] = upload of the queries, out[1] = the search kernel, out[2] = download of labels / distances / counts (milliseconds,
// HIP events on the default stream around the three steps).  SURVEY.md §8(d): "report H2D separately".
extern "C" int hnsw_gpu_last_batch_ms(hnsw_gpu_index *ix, float out[3])
{
	if (!ix || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	if (!ix->hb_valid || ix->ws.launches == 0) return fail(HNSW_GPU_ERR_ARG, "no host-pointer batch call (copy path) has completed on this mirror");
	HIPCHK(hipSetDevice(ix->device));
	const int evi = (int) ((ix->ws.launches - 1) % SearchWs::EV_RING);
	HIPCHK(hipEventSynchronize(ix->hb1));
	HIPCHK(hipEventElapsedTime(&out[0], ix->hb0, ix->ws.ev0[evi]));
	HIPCHK(hipEventElapsedTime(&out[1], ix->ws.ev0[evi], ix->ws.ev1[evi]));
	HIPCHK(hipEventElapsedTime(&out[2], ix->ws.ev1[evi], ix->hb1));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_search_kernel(hnsw_gpu_index *ix, char *buf, size_t len)
{
	if (!ix || !buf || len == 0) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	snprintf(buf, len, "%s", ix->ws.kname);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_abort(hnsw_gpu_index *ix)
{
	// no ix->mu here: the thread that holds it may be the one waiting for the launch this call is meant to end
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	std::lock_guard<std::mutex> g(g_ws_mu);
	return abort_ws_locked(&ix->ws) ? HNSW_GPU_OK : fail(HNSW_GPU_ERR_INTERNAL, "the workspace has no abort word");
}

extern "C" int hnsw_gpu_index_health(hnsw_gpu_index *ix, uint32_t *out8)
{
	if (!ix || !out8) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::recursive_mutex> g(ix->mu);
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipMemcpy(out8, ix->ws.health, 32, hipMemcpyDeviceToHost));
	out8[0] = __atomic_load_n(ix->ws.abort_host, __ATOMIC_SEQ_CST);
	out8[5] = __atomic_load_n(&ix->ws.abort_requests, __ATOMIC_SEQ_CST);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_search_slots(hnsw_gpu_index *ix, uint32_t *slots)
{
	if (!ix || !slots) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	*slots = ix->ws.last_slots;
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// search contexts: independent batches in flight on different streams
// ------------------------------------------------------------------------------------

extern "C" int hnsw_gpu_ctx_create(hnsw_gpu_index *ix, hnsw_gpu_ctx **out)
{
	if (!ix || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(ix->device));
	hnsw_gpu_ctx *c = new (std::nothrow) hnsw_gpu_ctx();
	if (!c) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	c->ix = ix;
	int rc = ws_init(&c->ws);
	if (rc) { ws_free(&c->ws); delete c; return rc; }
	*out = c;
	return HNSW_GPU_OK;
}

extern "C" void hnsw_gpu_ctx_destroy(hnsw_gpu_ctx *c)
{
	if (!c) return;
	(void) hipSetDevice(c->ix->device);
	ws_free(&c->ws);
	if (c->stage) (void) hipFree(c->stage);
	if (c->stream) (void) hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int hnsw_gpu_search_batch_ctx(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef,
										 label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
										 void *stream)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	return launch_search(c->ix, &c->ws, d_queries, c->ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats,
						 (hipStream_t) stream);
}

// 8-wave team blocks the device holds at once for rows wider than 320 floats (one per CU: 2 waves/SIMD): the figure a host sizes
// hnsw_gpu_ctx_set_walkers by.  <= 0: no such device.
extern "C" int hnsw_gpu_device_blocks(int device)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void) hipGetLastError(); return 0; }
	return prop.multiProcessorCount;
}

extern "C" int hnsw_gpu_ctx_set_walkers(hnsw_gpu_ctx *c, unsigned per_block)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	c->ws.walkers_hint = per_block;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_ctx_search_ms(hnsw_gpu_ctx *c, unsigned back, float *ms)
{
	if (!c || !ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return ws_search_ms(c->ix->device, &c->ws, back, ms);
}

// Host-pointer form of a context search: copy in, launch, copy out on the context's own stream and
// wait for that stream only, so host threads that own one context each keep several batches in
// flight on the device (the batching server's dispatchers, server_main.cpp).  Buffers from
// hnsw_gpu_host_alloc make the copies true DMA transfers.  One caller at a time per context.
extern "C" int hnsw_gpu_search_batch_ctx_host(hnsw_gpu_ctx *c, const coord_t *queries, size_t nq, size_t ef,
											  label_t *labels, dist_t *dists, uint32_t *counts)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	hnsw_gpu_index *ix = c->ix;
	HIPCHK(hipSetDevice(ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	const size_t dim = ix->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	if (qb + lb + db + cb > c->stage_bytes)
	{
		if (c->stage) (void) hipFree(c->stage);
		c->stage = nullptr; c->stage_bytes = 0;
		const size_t want = std::max<size_t>(qb + lb + db + cb, (size_t) 1 << 20);
		HIPCHK(hipMalloc(&c->stage, want));
		c->stage_bytes = want;
	}
	char *p = (char *) c->stage;
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	HIPCHK(hipMemcpyAsync(dq, queries, nq * dim * 4, hipMemcpyHostToDevice, c->stream));
	int rc = launch_search(ix, &c->ws, dq, dim, nq, ef, 0, dl, nullptr, dd, dc, nullptr, c->stream);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost, c->stream));
	if (dists) HIPCHK(hipMemcpyAsync(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(counts, dc, nq * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (size_t i = 0; i < nq; i++)                            // (as hnsw_gpu_search_batch: an interrupted launch is an error of a host-pointer call)
		if (counts[i] == ABORTED_COUNT)
			return fail(HNSW_GPU_ERR_INTERNAL, "the search launch was asked to end early (abort word): query %zu has no result", i);
	return HNSW_GPU_OK;
}

// Streamed completion (hnsw_gpu.h): device-pointer launch on the context's own stream with per-query
// completion flags.  Nothing is copied and nothing is waited for here.
extern "C" int hnsw_gpu_search_batch_ctx_flags(hnsw_gpu_ctx *c, const coord_t *d_queries, size_t nq, size_t ef,
											   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, uint32_t *d_stats,
											   uint32_t *d_done)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (!d_done) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	HIPCHK(hipSetDevice(c->ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	std::unique_lock<std::recursive_mutex> lock_(c->ix->mu);      // done_next -> launch is one step
	c->ws.done_next = d_done;
	int rc = launch_search(c->ix, &c->ws, d_queries, c->ix->meta.dim, nq, ef, 0, d_labels, nullptr, d_dists, d_counts, d_stats,
						   c->stream);
	c->ws.done_next = nullptr;
	return rc;
}

