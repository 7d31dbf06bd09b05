// gpu_build.hip — insert path (hnswalg.cpp:117-232, 279-291): batched link step (csrc/device_build.h), single inserts (csrc/device_insert.h)
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"
#include "device_build.h"
#include "device_insert.h"

// ------------------------------------------------------------------------------------
// insert path: link stored elements into the graph (device_build.h)
// ------------------------------------------------------------------------------------
extern "C" int pgemb_sort_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, int n, void *stream);

typedef void (*build_kernel_t)(const BuildArgs);

// ext_*: the candidates of ONE new element (count == 1) as a search already produced them — ascending by (dist, idx), the order
// searchBaseLayer's results leave hnsw_gpu_search_base* in — in memory the device can read (pinned host memory will do): the link
// step then runs without a search of its own (hnsw_gpu_index_insert_candidates).
static int link_range(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio, void *stream_,
					  const uint32_t *ext_idx, const float *ext_dist, const uint32_t *ext_cnt)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (first + count > ix->n) return fail(HNSW_GPU_ERR_ARG, "elements [%zu, %zu) are not stored (count %zu)", first, first + count, ix->n);
	if (count == 0) return HNSW_GPU_OK;
	if (ext_idx && (count != 1 || !ext_dist || !ext_cnt)) return fail(HNSW_GPU_ERR_ARG, "external candidates are for one element");
	if (max_batch == 0) max_batch = 4096;
	if (ratio == 0) ratio = 8;
	const size_t efc = ix->meta.efConstruction, M = ix->meta.M, maxM = ix->meta.maxM;
	if (efc == 0 || M == 0 || M > maxM) return fail(HNSW_GPU_ERR_ARG, "bad efConstruction/M");
	HIPCHK(hipSetDevice(ix->device));
	hipStream_t stream = (hipStream_t) stream_;

	// scratch carve
	max_batch = std::min(max_batch, count);
	const size_t slots = max_batch * M;
	if (slots >= 0x7FFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "batch too large");
	size_t tmp_bytes = 0;
	if (pgemb_sort_u64(nullptr, &tmp_bytes, nullptr, nullptr, (int) slots, stream) != 0)
		return fail(HNSW_GPU_ERR_HIP, "radix sort sizing failed");
	const size_t o_idx = 0;
	const size_t o_dist = o_idx + round_up(max_batch * efc * 4, 256);
	const size_t o_cnt = o_dist + round_up(max_batch * efc * 4, 256);
	const size_t o_pairs = o_cnt + round_up(max_batch * 4, 256);
	const size_t o_sorted = o_pairs + round_up(slots * 8, 256);
	const size_t o_seg = o_sorted + round_up(slots * 8, 256);
	const size_t o_ctr = o_seg + round_up(slots * 4, 256);
	const size_t o_tmp = o_ctr + 256;
	const size_t total = o_tmp + round_up(tmp_bytes, 256);
	if (max_batch > ix->bld_batch || tmp_bytes > ix->bld_tmp_bytes || !ix->bld)
	{
		if (ix->bld) (void) hipFree(ix->bld);
		ix->bld = nullptr; ix->bld_batch = 0;
		HIPCHK(hipMalloc(&ix->bld, total));
		ix->bld_batch = max_batch; ix->bld_tmp_bytes = tmp_bytes;
	}
	char *B = (char *) ix->bld;
	uint32_t *cand_idx = (uint32_t *) (B + o_idx);
	float *cand_dist = (float *) (B + o_dist);
	uint32_t *cand_cnt = (uint32_t *) (B + o_cnt);
	uint64_t *pairs = (uint64_t *) (B + o_pairs), *sorted = (uint64_t *) (B + o_sorted);
	uint32_t *seg = (uint32_t *) (B + o_seg), *ctr = (uint32_t *) (B + o_ctr);

	BuildArgs a;
	memset(&a, 0, sizeof(a));
	a.vec = ix->vec; a.links = ix->links;
	a.dim = (uint32_t) ix->meta.dim; a.stride = ix->stride; a.nchunks = ix->stride / 4; a.kiters = (a.nchunks + 15) / 16;
	a.qpad_floats = (uint32_t) round_up(a.kiters, BUILD_KB) * 64;
	a.maxM = (uint32_t) maxM; a.M = (uint32_t) M; a.lstride = ix->lstride; a.efc = (uint32_t) efc;
	a.cand_idx = cand_idx; a.cand_dist = cand_dist; a.cand_cnt = cand_cnt;
	a.pairs = pairs; a.npairs = ctr; a.sorted_pairs = sorted; a.seg_start = seg; a.nseg = ctr + 1; a.ticket = ctr + 2;
	const uint32_t cap = (uint32_t) round_up(std::max<size_t>(std::max(efc, maxM + 1), 128), 8);   // tmpd holds 2 x 64 sums
	a.wave_bytes = (uint32_t) round_up((size_t) a.qpad_floats * 4 + (size_t) cap * (8 * 2 + 4 * 3) + (maxM + 2) * 4, 16);
	if (a.wave_bytes > LDS_PER_CU) return fail(HNSW_GPU_ERR_ARG, "efConstruction/maxM/dim need too much LDS (%u bytes)", a.wave_bytes);
	uint32_t wpb = 4;
	while (wpb > 1 && (size_t) wpb * a.wave_bytes > 64 * 1024) wpb >>= 1;
	const size_t lds = (size_t) wpb * a.wave_bytes;
	build_kernel_t ksel, krev;
	switch ((int) ix->meta.dist_func)
	{
		case F_L2:     ksel = select_links_kernel<F_L2>;        krev = reverse_links_kernel<F_L2>; break;
		case F_COSINE: ksel = select_links_kernel<F_COSINE>;    krev = reverse_links_kernel<F_COSINE>; break;
		default:       ksel = select_links_kernel<F_MANHATTAN>; krev = reverse_links_kernel<F_MANHATTAN>; break;
	}
	if (lds > 48 * 1024)
	{
		HIPCHK(hipFuncSetAttribute((const void *) ksel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		HIPCHK(hipFuncSetAttribute((const void *) krev, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	}

	size_t linked = first, end = first + count;
	if (linked == 0) linked = 1;                    // element 0 is never bound: hnswalg.cpp:228
	while (linked < end)
	{
		const size_t b = std::min({end - linked, max_batch, std::max<size_t>(1, linked / ratio)});
		const size_t bslots = b * M;
		const bool single = b == 1;                     // the reference's serial insert: no sort, no segment marking (device_build.h)
		HIPCHK(hipMemsetAsync(ctr, 0, 16, stream));
		if (!single) HIPCHK(hipMemsetAsync(pairs, 0xFF, bslots * 8, stream));
		// 1. searchBaseLayer(ef = efConstruction) for every new element (hnswalg.cpp:229) — unless the caller brought its result
		if (ext_idx)
		{
			a.cand_idx = ext_idx; a.cand_dist = ext_dist; a.cand_cnt = ext_cnt;
		}
		else
		{
			a.cand_idx = cand_idx; a.cand_dist = cand_dist; a.cand_cnt = cand_cnt;
			int rc = launch_search(ix, &ix->ws, ix->vec + linked * ix->stride, ix->stride, b, efc, 1, nullptr, cand_idx, cand_dist,
								   cand_cnt, nullptr, stream);
			if (rc) return rc;
		}
		// 2. choose links, emit reverse pairs
		a.first = (uint32_t) linked; a.count = (uint32_t) b; a.pair_slots = (uint32_t) bslots;
		a.single = single ? 1u : 0u; a.seg_out = seg; a.nseg_out = ctr + 1;
		a.sorted_pairs = single ? pairs : sorted;
		hipLaunchKernelGGL(ksel, dim3((uint32_t) ((b + wpb - 1) / wpb)), dim3(wpb * 64), lds, stream, a);
		// 3. reverse edges grouped by target
		if (!single)
		{
			size_t tb = ix->bld_tmp_bytes;
			if (pgemb_sort_u64(B + o_tmp, &tb, pairs, sorted, (int) bslots, stream) != 0)
				return fail(HNSW_GPU_ERR_HIP, "radix sort failed");
			hipLaunchKernelGGL(mark_segments_kernel, dim3((uint32_t) ((bslots + 255) / 256)), dim3(256), 0, stream, sorted,
							   (uint32_t) bslots, seg, ctr + 1);
		}
		const uint32_t rblocks = (uint32_t) std::min<size_t>((bslots + wpb - 1) / wpb, (size_t) ix->num_cu * 4);
		hipLaunchKernelGGL(krev, dim3(rblocks), dim3(wpb * 64), lds, stream, a);
		HIPCHK(hipGetLastError());
		linked += b;
	}
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_link(hnsw_gpu_index *ix, size_t first, size_t count, size_t max_batch, size_t ratio,
								   void *stream_)
{
	return link_range(ix, first, count, max_batch, ratio, stream_, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------
// small accessors used by the drop-in insert (embedding_shim.cpp)
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_index_get_links(hnsw_gpu_index *ix, idx_t idx, idx_t *out)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !out || idx >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx);
	HIPCHK(hipSetDevice(ix->device));
	const size_t maxM = ix->meta.maxM;
	uint32_t tmp[4096 + 16];
	HIPCHK(hipMemcpy(tmp, ix->links + (size_t) idx * ix->lstride, ix->lstride * 4, hipMemcpyDeviceToHost));
	uint32_t cnt = 0;
	for (size_t j = 0; j < maxM; j++)
		if (tmp[j] != LINK_NONE) out[1 + cnt++] = tmp[j];
	out[0] = cnt;
	for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	return HNSW_GPU_OK;
}

// The link list of one element and the lists of all its neighbours in one launch + one wait: what an insert changed
// (hnswalg.cpp:169-222: the new element's list and a reverse link in each neighbour's), for the write-back of
// hnsw_bind_point.  Rows land in the mirror's pinned staging; block 0 = the element, block 1+j = its j-th link slot.
// done_ctr / flag (hnsw_gpu_index_insert_*): the block that finishes LAST stores the completion flag behind a system-scope release —
// the lists of every block are in host memory before the flag, and no extra launch is needed for it.
__global__ __launch_bounds__(64) void gather_link_lists_kernel(const uint32_t *__restrict__ links, uint32_t lstride, uint32_t idx,
															   uint32_t n, uint32_t *__restrict__ out, uint32_t *done_ctr, uint32_t *flag)
{
	uint32_t src = idx;
	bool have = true;
	if (blockIdx.x > 0)
	{
		src = links[(size_t) idx * lstride + (blockIdx.x - 1)];
		have = src != LINK_NONE && src < n;
	}
	for (uint32_t j = threadIdx.x; j < lstride; j += 64)
		out[(size_t) blockIdx.x * lstride + j] = have ? links[(size_t) src * lstride + j] : LINK_NONE;
	if (flag)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: this block's rows
		uint32_t last = 0;
		if (threadIdx.x == 0) last = atomicAdd(done_ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
		if (__builtin_amdgcn_readfirstlane(last))
		{
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");    // every other block's release happened before its increment
			if (threadIdx.x == 0)
			{
				atomicExch(done_ctr, 0u);                     // ready for the next insert: no memset between calls
				__hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
}

extern "C" int hnsw_gpu_index_get_link_lists(hnsw_gpu_index *ix, idx_t idx, idx_t *mine, idx_t *others)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !mine || !others || idx >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx);
	HIPCHK(hipSetDevice(ix->device));
	const size_t maxM = ix->meta.maxM, ls = ix->lstride;
	const size_t need = (maxM + 1) * ls * 4;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, std::max<size_t>(need, 64 << 10), hipHostMallocDefault));
		ix->pin_bytes = std::max<size_t>(need, 64 << 10);
	}
	uint32_t *h = (uint32_t *) ix->pin;
	hipLaunchKernelGGL(gather_link_lists_kernel, dim3((uint32_t) maxM + 1), dim3(64), 0, 0, ix->links, (uint32_t) ls, (uint32_t) idx,
					   (uint32_t) ix->n, h, (uint32_t *) nullptr, (uint32_t *) nullptr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(nullptr));
	auto compact = [&](const uint32_t *row, idx_t *out)
	{
		uint32_t cnt = 0;
		for (size_t j = 0; j < maxM; j++)
			if (row[j] != LINK_NONE) out[1 + cnt++] = row[j];
		out[0] = cnt;
		for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	};
	compact(h, mine);
	// neighbour j of the compacted list sits in link slot s_j of the row (slots may hold holes)
	size_t k = 0;
	for (size_t s = 0; s < maxM && k < mine[0]; s++)
		if (h[s] != LINK_NONE) { compact(h + (1 + s) * ls, others + k * (maxM + 1)); k++; }
	return HNSW_GPU_OK;
}

// Gathered link rows (row 0 = the element, row 1 + s = the element in its link slot s) -> the compacted [count | links] lists of
// the element and of each neighbour, in the order of the compacted list.
static void compact_lists(const uint32_t *lists, size_t maxM, size_t ls, idx_t *mine, idx_t *others)
{
	auto compact = [&](const uint32_t *row, idx_t *out)
	{
		uint32_t cnt = 0;
		for (size_t j = 0; j < maxM; j++)
			if (row[j] != LINK_NONE) out[1 + cnt++] = row[j];
		out[0] = cnt;
		for (size_t j = cnt; j < maxM; j++) out[1 + j] = 0;
	};
	compact(lists, mine);
	size_t k = 0;
	for (size_t s2 = 0; s2 < maxM && k < mine[0]; s2++)
		if (lists[s2] != LINK_NONE) { compact(lists + (1 + s2) * ls, others + k * (maxM + 1)); k++; }
}

typedef void (*insert_kernel_t)(const InsertArgs);

static std::atomic<uint64_t> g_inserts_two_launch{0}, g_inserts_general{0};
extern "C" void hnsw_gpu_insert_path_counts(uint64_t out[2])
{
	if (!out) return;
	out[0] = g_inserts_two_launch.load();
	out[1] = g_inserts_general.load();
}

static bool insert_fused_wanted()
{
	knobs_init();
	return knob(K_INSERT_FUSED, 1) != 0;                     // 0: the four-launch path of round 3's first half (A/B runs, tests of both)
}

// The two-launch insert's shape for this mirror: *lds = 0 when max(efConstruction, maxM + 1) candidates are more than the chain
// of device_insert.h keeps in one wavefront's registers (INS_MAX_SIDE) or a block does not fit a CU's LDS — the caller then takes
// the general builder path.  Device scratch (ix->ins): candidates of the insert's own walk | targets | bit matrix.
static int plan_insert(hnsw_gpu_index *ix, InsertArgs *a, size_t *lds)
{
	*lds = 0;
	memset(a, 0, sizeof(*a));
	const size_t efc = ix->meta.efConstruction, M = ix->meta.M, maxM = ix->meta.maxM;
	if (efc == 0 || M == 0 || M > maxM) return fail(HNSW_GPU_ERR_ARG, "bad efConstruction/M");
	BuildArgs &b = a->b;
	b.vec = ix->vec; b.links = ix->links;
	b.dim = (uint32_t) ix->meta.dim; b.stride = ix->stride; b.nchunks = ix->stride / 4; b.kiters = (b.nchunks + 15) / 16;
	b.qpad_floats = (uint32_t) round_up(b.kiters, INS_KB) * 64;
	b.maxM = (uint32_t) maxM; b.M = (uint32_t) M; b.lstride = ix->lstride; b.efc = (uint32_t) efc;
	const size_t side = round_up(std::max(efc, maxM + 1), 64);
	if (side > INS_MAX_SIDE) return HNSW_GPU_OK;
	const size_t cap = side;
	const size_t shared = cap * 8 * 3 + side * (side / 64) * 8 + round_up(maxM + 2, 4) * 4 + 16;
	const size_t per_wave = ((size_t) b.qpad_floats + 128) * 4;
	const size_t lds_limit = std::min(LDS_PER_CU, ix->max_lds);                     // what ONE block may ask for on this device
	if (lds_limit < 2048) return HNSW_GPU_OK;
	size_t nw = 8;
	while (nw > 1 && shared + nw * per_wave > lds_limit - 1024) nw >>= 1;
	if (shared + nw * per_wave > lds_limit - 1024) return HNSW_GPU_OK;
	size_t nw2 = 12;                                                                // step 2: up to 12 wavefronts around one target (device_insert.h)
	while (nw2 > nw && shared + nw2 * per_wave > lds_limit - 1024) nw2 -= 4;
	if (nw2 < nw) nw2 = nw;
	a->nw = (uint32_t) nw; a->nw2 = (uint32_t) nw2; a->side = (uint32_t) side; a->cap = (uint32_t) cap;
	const size_t o_ci = 0, o_cd = o_ci + round_up(efc * 4, 256), o_cc = o_cd + round_up(efc * 4, 256);
	const size_t o_tg = o_cc + 256, o_bits = o_tg + round_up(M * 4, 256), total = o_bits + side * (side / 16) * 2;
	if (ix->ins_bytes < total)
	{
		if (ix->ins) (void) hipFree(ix->ins);
		ix->ins = nullptr; ix->ins_bytes = 0;
		HIPCHK(hipMalloc(&ix->ins, total));
		ix->ins_bytes = total;
	}
	char *S = (char *) ix->ins;
	b.cand_idx = (const uint32_t *) (S + o_ci); b.cand_dist = (const float *) (S + o_cd); b.cand_cnt = (const uint32_t *) (S + o_cc);
	a->targets = (uint32_t *) (S + o_tg); a->bits = (uint16_t *) (S + o_bits);
	a->labels = ix->labels;
	a->ntargets = ix->misc + 5; a->done1 = ix->misc + 4; a->done2 = ix->misc + 3;
	*lds = shared + nw2 * per_wave;                                                 // (the larger of the two carves: one attribute for both kernels)
	return HNSW_GPU_OK;
}

// hnsw_bind_point's device side in ONE host call (hnswalg.cpp:279-291, 225-232): element `idx` (= the mirror's current count)
// is appended and linked exactly as the reference's serial insert links it, and the changed link lists — its own and one per
// selected neighbour — come back compacted ([count | links], maxM + 1 words each) for the host's write-back.  Everything is
// enqueued on the default stream without a host wait in between: the row and its label are read by the append kernel straight
// from pinned host memory, the gathered lists are written straight into it, and a one-thread kernel behind them stores a
// completion flag that the calling core polls (no copy engine, no interrupt wake-up: the few-queries mechanics of
// hnsw_gpu_search_batch).  Round 2 made this call as append (2 blocking copies + sync) + link (2 memsets, search, select, a
// hipCUB radix sort, segment marking, reverse) + get_link_lists (launch + sync): 0.85-1.5 ms per row against the reference's
// 0.06-0.12 ms; a single row needs no sort (its neighbours are distinct targets) and no waits.
static int insert_impl(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx, const dist_t *cand_dist,
					   uint32_t ncand, idx_t *mine, idx_t *others)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !point || !mine || !others) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (cand_idx && (!cand_dist || ncand > ix->meta.efConstruction)) return fail(HNSW_GPU_ERR_ARG, "bad candidate list");
	if (cand_idx)
	{
		// the kernels use these numbers as row and link addresses and rely on searchBaseLayer's order: stored elements only,
		// strictly ascending by (dist, idx) — which also makes them distinct (at most 512 entries: nothing next to the insert)
		for (uint32_t i = 0; i < ncand; i++)
		{
			if (cand_idx[i] >= idx) return fail(HNSW_GPU_ERR_ARG, "candidate %u is element %u, not below the new element %u", i, (unsigned) cand_idx[i], (unsigned) idx);
			if (cand_dist[i] != cand_dist[i]) return fail(HNSW_GPU_ERR_ARG, "candidate %u has a NaN distance", i);
			if (i > 0 && !(cand_dist[i - 1] < cand_dist[i] || (cand_dist[i - 1] == cand_dist[i] && cand_idx[i - 1] < cand_idx[i])))
				return fail(HNSW_GPU_ERR_ARG, "candidates %u and %u are not in ascending (dist, idx) order", i - 1, i);
		}
	}
	if ((size_t) idx != ix->n) return fail(HNSW_GPU_ERR_ARG, "insert_one(%u): the mirror holds %zu elements", (unsigned) idx, ix->n);
	if (ix->n + 1 > ix->cap) return fail(HNSW_GPU_ERR_ARG, "insert exceeds capacity (%zu)", ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	if (ix->trace_active) { HIPCHK(hipStreamSynchronize(nullptr)); ix->trace_active = false; }   // an abandoned trace still writes the staging
	if (ix->ins_dirty)
	{
		// a previous insert failed after its kernels were enqueued: the last-block counters (misc words 3..5) may be non-zero, and
		// with them every later insert would mis-detect its last block and never store the completion flag
		HIPCHK(hipDeviceSynchronize());
		HIPCHK(hipMemset(ix->misc + 3, 0, 12));
		ix->ins_dirty = false;
	}
	const size_t dim = ix->meta.dim, maxM = ix->meta.maxM, ls = ix->lstride;
	const size_t efc_ = ix->meta.efConstruction;
	const size_t o_lab = round_up(dim * 4, 8), o_lists = round_up(o_lab + 8, 256), o_flag = o_lists + round_up((maxM + 1) * ls * 4, 256);
	const size_t o_ci = o_flag + 256, o_cd = o_ci + round_up(efc_ * 4, 256), o_cc = o_cd + round_up(efc_ * 4, 256);
	const size_t need = o_cc + 256;
	if (ix->pin_bytes < need)
	{
		if (ix->pin) (void) hipHostFree(ix->pin);
		ix->pin = nullptr; ix->pin_bytes = 0;
		HIPCHK(hipHostMalloc((void **) &ix->pin, std::max<size_t>(need, 64 << 10), hipHostMallocDefault));
		ix->pin_bytes = std::max<size_t>(need, 64 << 10);
	}
	char *h = ix->pin;
	memcpy(h, point, dim * 4);
	memcpy(h + o_lab, &label, 8);
	volatile uint32_t *flag = (volatile uint32_t *) (h + o_flag);
	*flag = 0;
	uint32_t *lists = (uint32_t *) (h + o_lists);
	int rc;
	InsertArgs ia;
	size_t ilds = 0;
	insert_kernel_t ksel = nullptr, krev = nullptr;
	bool two_launch = insert_fused_wanted() && plan_insert(ix, &ia, &ilds) == HNSW_GPU_OK && ilds;
	if (two_launch)
	{
		switch ((int) ix->meta.dist_func)
		{
			case F_L2:     ksel = insert_select_kernel<F_L2>;        krev = insert_reverse_kernel<F_L2>; break;
			case F_COSINE: ksel = insert_select_kernel<F_COSINE>;    krev = insert_reverse_kernel<F_COSINE>; break;
			default:       ksel = insert_select_kernel<F_MANHATTAN>; krev = insert_reverse_kernel<F_MANHATTAN>; break;
		}
		static std::atomic<size_t> lds_allowed[3][8];          // per function and device: the attribute is set when a larger carve comes along, not per insert
		std::atomic<size_t> &allowed = lds_allowed[std::min(std::max((int) ix->meta.dist_func, 0), 2)][ix->device & 7];
		if (ilds > 48 * 1024 && (ilds > allowed.load() || ix->device > 7))
		{
			// a device that refuses the carve takes the general builder path (plan_insert's contract), it does not fail the insert
			if (hipFuncSetAttribute((const void *) ksel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ilds) != hipSuccess ||
				hipFuncSetAttribute((const void *) krev, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ilds) != hipSuccess)
			{
				(void) hipGetLastError();
				two_launch = false;
			}
			else allowed.store(ilds);
		}
	}
	if (two_launch)
	{
		// two launches (device_insert.h): [append +] pair triangle + chain | one block per target + the flag
		ia.b.first = (uint32_t) idx; ia.b.count = 1;
		ia.bind = idx > 0 ? 1u : 0u;
		ia.lists_out = lists; ia.flag = (uint32_t *) (h + o_flag);
		if (cand_idx || idx == 0)                            // the walk has been done (and validated) already: the kernel reads its result, the row and the label from pinned memory
		{
			if (cand_idx)
			{
				memcpy(h + o_ci, cand_idx, (size_t) ncand * 4);
				memcpy(h + o_cd, cand_dist, (size_t) ncand * 4);
			}
			*(uint32_t *) (h + o_cc) = ncand;
			ia.ncand_p1 = ncand + 1;                         // known here: the kernels do not fetch it over the bus
			ia.b.cand_idx = (const uint32_t *) (h + o_ci); ia.b.cand_dist = (const float *) (h + o_cd); ia.b.cand_cnt = (const uint32_t *) (h + o_cc);
			ia.src_row = (const float *) h; ia.src_label = (const uint64_t *) (h + o_lab);
			ix->n += 1;                                      // stored by step 1's block 0
		}
		else                                                 // searchBaseLayer(ef = efConstruction) of the insert itself (hnswalg.cpp:229), the point read as
		{                                                    // the query straight from pinned memory over the idx elements stored so far; step 1 stores the row
			rc = launch_search(ix, &ix->ws, (const float *) h, dim, 1, efc_, 1, nullptr, (uint32_t *) ia.b.cand_idx,
							   (float *) ia.b.cand_dist, (uint32_t *) ia.b.cand_cnt, nullptr, nullptr);
			if (rc) return rc;
			ia.src_row = (const float *) h; ia.src_label = (const uint64_t *) (h + o_lab);
			ix->n += 1;
		}
		// (one wavefront per unit of the bit triangle: device_insert.h)
		const uint32_t g1 = std::max<uint32_t>(1u, (units_for((uint32_t) efc_) + ia.nw - 1) / ia.nw);
		hipLaunchKernelGGL(ksel, dim3(g1), dim3(ia.nw * 64), ilds, 0, ia);
		if (hipError_t le = hipGetLastError(); le != hipSuccess)
		{
			ix->n = idx;                                     // nothing was stored
			return fail(HNSW_GPU_ERR_HIP, "insert step 1 did not launch: %s", hipGetErrorString(le));
		}
		hipLaunchKernelGGL(krev, dim3((uint32_t) ix->meta.M), dim3(ia.nw2 * 64), ilds, 0, ia);
		if (hipError_t le = hipGetLastError(); le != hipSuccess)
		{
			// step 1 runs (row, label, own list), step 2 never will: no element points at the new one, so the mirror without it is
			// the mirror before the call; step 1's last block has reset its own counter, the next insert re-checks all three
			(void) hipStreamSynchronize(nullptr);
			ix->n = idx;
			ix->ins_dirty = true;
			return fail(HNSW_GPU_ERR_HIP, "insert step 2 did not launch: %s", hipGetErrorString(le));
		}
		rc = poll_done_flag(flag, "an insert", nullptr);
		if (rc)
		{
			// the element is stored and (partly) linked: the graph is searchable but not the reference's; the caller sees the error
			// and re-mirrors (embedding_shim.cpp drops its mirror on any insert error)
			ix->ins_dirty = true;
			return rc;
		}
		compact_lists(lists, maxM, ls, mine, others);
		g_inserts_two_launch++;
		return HNSW_GPU_OK;
	}
	g_inserts_general++;
	rc = hnsw_gpu_index_append_dev(ix, (const coord_t *) h, (const label_t *) (h + o_lab), 1, nullptr);
	if (rc) return rc;
	if (idx > 0)                                             // element 0 is never bound (hnswalg.cpp:228)
	{
		if (cand_idx)                                        // the walk has been done (and validated) already: its result, from pinned memory
		{
			memcpy(h + o_ci, cand_idx, (size_t) ncand * 4);
			memcpy(h + o_cd, cand_dist, (size_t) ncand * 4);
			*(uint32_t *) (h + o_cc) = ncand;
			rc = link_range(ix, idx, 1, 1, 0, nullptr, (const uint32_t *) (h + o_ci), (const float *) (h + o_cd), (const uint32_t *) (h + o_cc));
		}
		else
			rc = link_range(ix, idx, 1, 1, 0, nullptr, nullptr, nullptr, nullptr);
		if (rc) return rc;
	}
	// (the gather's last block stores the flag; its block counter is word 3 of the mirror's misc words: zero at creation, reset by that block)
	hipLaunchKernelGGL(gather_link_lists_kernel, dim3((uint32_t) maxM + 1), dim3(64), 0, 0, ix->links, (uint32_t) ls, (uint32_t) idx,
					   (uint32_t) ix->n, lists, ix->misc + 3, (uint32_t *) (h + o_flag));
	HIPCHK(hipGetLastError());
	rc = poll_done_flag(flag, "an insert", nullptr);
	if (rc) { ix->ins_dirty = true; return rc; }
	compact_lists(lists, maxM, ls, mine, others);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_insert_one(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, idx_t *mine, idx_t *others)
{
	return insert_impl(ix, point, label, idx, nullptr, nullptr, 0, mine, others);
}

// The same with the candidate list given: what searchBaseLayer(point, ef = efConstruction) returned on THIS mirror a moment ago
// (hnsw_gpu_search_trace in base mode: element numbers and distances ascending by (dist, idx)) — a caller that has just walked for
// the point (the validated cache of the unmodified glue walks to CHECK the mirror, shim_cache.h) does not pay for the walk twice.
extern "C" int hnsw_gpu_index_insert_candidates(hnsw_gpu_index *ix, const coord_t *point, label_t label, idx_t idx, const idx_t *cand_idx,
												const dist_t *cand_dist, uint32_t ncand, idx_t *mine, idx_t *others)
{
	if (!cand_idx || !cand_dist) return fail(HNSW_GPU_ERR_ARG, "NULL candidate list");
	return insert_impl(ix, point, label, idx, cand_idx, cand_dist, ncand, mine, others);
}

