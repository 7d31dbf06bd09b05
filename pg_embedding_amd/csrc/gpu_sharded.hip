// gpu_sharded.hip — the exchange step of a row-sharded index: top-k merge, shards of ONE process, the buffer shared between processes
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"

// ------------------------------------------------------------------------------------
// multi-shard merge: nlists x (dist,label) lists per query -> ef best by (dist, label)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool dl_less(uint32_t da, uint64_t la, uint32_t db, uint64_t lb)
{
	return da < db || (da == db && la < lb);
}

__global__ __launch_bounds__(64) void topk_merge_kernel(const uint64_t *__restrict__ in_labels, const float *__restrict__ in_dists,
														size_t lstep, size_t dstep,       /* list-to-list strides, in elements */
														uint32_t nlists, uint32_t nq, uint32_t ef,
														uint64_t *__restrict__ out_labels, float *__restrict__ out_dists,
														uint32_t *__restrict__ out_counts)
{
	const uint32_t qi = blockIdx.x;
	const int lane = threadIdx.x;
	const uint32_t total = nlists * ef;
	uint32_t kept = 0;
	for (uint32_t x0 = 0; x0 < total; x0 += 64)
	{
		const uint32_t x = x0 + lane;
		bool emit = false;
		if (x < total)
		{
			const uint32_t l = x / ef, i = x - l * ef;
			const size_t at = (size_t) qi * ef + i;
			const uint64_t lab = in_labels[l * lstep + at];
			const uint32_t d = ord_f32(in_dists[l * dstep + at]);
			if (lab != ~0ull)
			{
				uint32_t rank = i;
				for (uint32_t m = 0; m < nlists && rank < ef; m++)
				{
					if (m == l) continue;
					const size_t ob = (size_t) qi * ef;
					uint32_t lo = 0, hi = ef;
					while (lo < hi)
					{
						const uint32_t mid = (lo + hi) >> 1;
						const uint64_t ol = in_labels[m * lstep + ob + mid];
						const uint32_t od = ord_f32(in_dists[m * dstep + ob + mid]);
						// equal keys (cannot happen for disjoint shards) go to the lower list number
						const bool below = (ol != ~0ull) && (dl_less(od, ol, d, lab) || (od == d && ol == lab && m < l));
						if (below) lo = mid + 1; else hi = mid;
					}
					rank += lo;
				}
				if (rank < ef)
				{
					out_labels[(size_t) qi * ef + rank] = lab;
					if (out_dists) out_dists[(size_t) qi * ef + rank] = unord_f32(d);
					emit = true;
				}
			}
		}
		kept += (uint32_t) __builtin_popcountll(__ballot(emit));
	}
	for (uint32_t i = kept + lane; i < ef; i += 64)
	{
		out_labels[(size_t) qi * ef + i] = ~0ull;
		if (out_dists) out_dists[(size_t) qi * ef + i] = __builtin_inff();
	}
	if (lane == 0) out_counts[qi] = kept;
}

extern "C" int hnsw_gpu_merge_topk_strided_dev(int device, const label_t *d_in_labels, size_t label_list_stride,
											   const dist_t *d_in_dists, size_t dist_list_stride, size_t nlists,
											   size_t nq, size_t ef, label_t *d_out_labels, dist_t *d_out_dists,
											   uint32_t *d_out_counts, void *stream)
{
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_in_labels || !d_in_dists || !d_out_labels || !d_out_counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (nlists == 0 || ef == 0) return fail(HNSW_GPU_ERR_ARG, "nlists and ef must be positive");
	if (nlists * ef >= 0xFFFFFFFFull || nq >= 0x7FFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "merge too large");
	if (label_list_stride < nq * ef || dist_list_stride < nq * ef) return fail(HNSW_GPU_ERR_ARG, "list stride smaller than one list");
	HIPCHK(hipSetDevice(device));
	hipLaunchKernelGGL(topk_merge_kernel, dim3((uint32_t) nq), dim3(64), 0, (hipStream_t) stream, d_in_labels, d_in_dists,
					   label_list_stride, dist_list_stride, (uint32_t) nlists, (uint32_t) nq, (uint32_t) ef, d_out_labels, d_out_dists,
					   d_out_counts);
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_merge_topk_dev(int device, const label_t *d_in_labels, const dist_t *d_in_dists, size_t nlists,
									   size_t nq, size_t ef, label_t *d_out_labels, dist_t *d_out_dists,
									   uint32_t *d_out_counts, void *stream)
{
	return hnsw_gpu_merge_topk_strided_dev(device, d_in_labels, nq * ef, d_in_dists, nq * ef, nlists, nq, ef, d_out_labels,
										   d_out_dists, d_out_counts, stream);
}

// ------------------------------------------------------------------------------------
// A device buffer shared between PROCESSES: the exchange buffer of a row-sharded search whose shards live in different
// processes (one GPU-owning server per GPU).  Every process searches its shard with its output pointers inside the buffer
// (hnsw_gpu_search_batch_dev: slot r = the r-th [nq][ef] block), the owner merges (hnsw_gpu_merge_topk_strided_dev) once the
// others have told it — over whatever channel they already share — that their launches are complete.  With the importer on
// another GPU its stores cross xGMI as peer stores, exactly like the one-process form (hnsw_gpu_sharded_search_dev); no
// staging copy, no collective library in a C host.
// ------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_device_wait(int device, void *stream)
{
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipStreamSynchronize((hipStream_t) stream));
	return HNSW_GPU_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(hnsw_gpu_ipc_handle), "the ABI's handle must hold a HIP IPC handle");

extern "C" int hnsw_gpu_shared_alloc(int device, size_t bytes, void **d_ptr, hnsw_gpu_ipc_handle *handle)
{
	if (!d_ptr || !handle || bytes == 0) return fail(HNSW_GPU_ERR_ARG, "NULL argument or empty buffer");
	HIPCHK(hipSetDevice(device));
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) { (void) hipGetLastError(); return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
	hipIpcMemHandle_t h;
	e = hipIpcGetMemHandle(&h, p);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		(void) hipFree(p);
		return fail(HNSW_GPU_ERR_HIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set where the driver only has dmabuf IPC?)", hipGetErrorString(e));
	}
	memset(handle, 0, sizeof(*handle));
	memcpy(handle->bytes, &h, sizeof(h));
	*d_ptr = p;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_open(int device, const hnsw_gpu_ipc_handle *handle, void **d_ptr)
{
	if (!d_ptr || !handle) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(device));
	hipIpcMemHandle_t h;
	memcpy(&h, handle->bytes, sizeof(h));
	void *p = nullptr;
	const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
	if (e != hipSuccess) { (void) hipGetLastError(); return fail(HNSW_GPU_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
	*d_ptr = p;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_close(int device, void *d_ptr)
{
	if (!d_ptr) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipIpcCloseMemHandle(d_ptr));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shared_free(int device, void *d_ptr)
{
	if (!d_ptr) return HNSW_GPU_OK;
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipFree(d_ptr));
	return HNSW_GPU_OK;
}

// ------------------------------------------------------------------------------------
// row-sharded index inside ONE process: shards on one or several devices, per-shard searchKnn,
// results written straight into the merge device's memory (peer access over xGMI), one merge kernel
// ------------------------------------------------------------------------------------
struct hnsw_gpu_sharded
{
	std::mutex mu;
	std::vector<hnsw_gpu_index *> shards;
	int home = 0;                                   // device of shard 0: queries arrive and results leave there
	std::vector<hipStream_t> streams;               // one per shard, on the shard's device
	std::vector<SearchWs *> ws;                     // ... and a search workspace of its own per shard: a direct search on a shard
	                                                // (its default workspace) and a sharded call never share tickets or bitmaps
	std::vector<hipEvent_t> done;                   // shard i's results are in the home device's gather buffer (timing enabled)
	hipEvent_t merge_start = nullptr;               // home: every shard's `done` has been waited for, the merge kernel is next
	bool timed = false;                             // a call has completed its enqueue: hnsw_gpu_sharded_last_ms has something to read
	std::vector<bool> direct;                       // the shard's device writes home memory directly
	std::vector<float *> q_local; std::vector<size_t> q_cap;          // query copy on a remote shard's device
	std::vector<char *> out_local; std::vector<size_t> out_cap;       // result block when not `direct`
	hipEvent_t ready = nullptr;
	hipEvent_t merged = nullptr; bool merged_set = false;             // end of the previous call's merge: `gather` may be rewritten after it
	char *gather = nullptr; size_t gather_bytes = 0;                  // home: nshards result blocks
	char *io = nullptr; size_t io_bytes = 0;                          // home: staging of the host-pointer form
	hipStream_t home_stream = nullptr;
};

extern "C" void hnsw_gpu_sharded_destroy(hnsw_gpu_sharded *s)
{
	if (!s) return;
	for (size_t i = 0; i < s->shards.size(); i++)
	{
		(void) hipSetDevice(s->shards[i]->device);
		if (i < s->streams.size() && s->streams[i]) (void) hipStreamDestroy(s->streams[i]);
		if (i < s->ws.size() && s->ws[i]) { ws_free(s->ws[i]); delete s->ws[i]; }
		if (i < s->done.size() && s->done[i]) (void) hipEventDestroy(s->done[i]);
		if (i < s->q_local.size() && s->q_local[i]) (void) hipFree(s->q_local[i]);
		if (i < s->out_local.size() && s->out_local[i]) (void) hipFree(s->out_local[i]);
	}
	(void) hipSetDevice(s->home);
	if (s->ready) (void) hipEventDestroy(s->ready);
	if (s->merged) (void) hipEventDestroy(s->merged);
	if (s->merge_start) (void) hipEventDestroy(s->merge_start);
	if (s->gather) (void) hipFree(s->gather);
	if (s->io) (void) hipFree(s->io);
	if (s->home_stream) (void) hipStreamDestroy(s->home_stream);
	delete s;
}

extern "C" int hnsw_gpu_sharded_create(hnsw_gpu_index *const *shards, size_t nshards, hnsw_gpu_sharded **out)
{
	if (!shards || !out || nshards == 0) return fail(HNSW_GPU_ERR_ARG, "need at least one shard");
	for (size_t i = 0; i < nshards; i++)
	{
		if (!shards[i]) return fail(HNSW_GPU_ERR_ARG, "shard %zu is NULL", i);
		if (shards[i]->meta.dim != shards[0]->meta.dim || shards[i]->meta.dist_func != shards[0]->meta.dist_func)
			return fail(HNSW_GPU_ERR_ARG, "shard %zu differs in dims / metric from shard 0", i);
	}
	hnsw_gpu_sharded *s = new (std::nothrow) hnsw_gpu_sharded();
	if (!s) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	s->shards.assign(shards, shards + nshards);
	s->home = shards[0]->device;
	s->streams.assign(nshards, nullptr); s->done.assign(nshards, nullptr); s->direct.assign(nshards, false);
	s->ws.assign(nshards, nullptr);
	s->q_local.assign(nshards, nullptr); s->q_cap.assign(nshards, 0);
	s->out_local.assign(nshards, nullptr); s->out_cap.assign(nshards, 0);
	hipError_t e = hipSuccess;
	for (size_t i = 0; i < nshards && e == hipSuccess; i++)
	{
		const int dev = shards[i]->device;
		if ((e = hipSetDevice(dev)) != hipSuccess) break;
		if ((e = hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking)) != hipSuccess) break;
		if ((e = hipEventCreate(&s->done[i])) != hipSuccess) break;
		s->ws[i] = new (std::nothrow) SearchWs();
		if (!s->ws[i] || ws_init(s->ws[i]) != HNSW_GPU_OK) { e = hipErrorOutOfMemory; break; }
		if (dev == s->home) s->direct[i] = true;
		else
		{
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, dev, s->home) == hipSuccess && can && (knobs_init(), knob(K_SHARDED_NO_PEER, 0) == 0))
			{
				const hipError_t pe = hipDeviceEnablePeerAccess(s->home, 0);
				if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) s->direct[i] = true;
				(void) hipGetLastError();
			}
		}
	}
	if (e == hipSuccess) e = hipSetDevice(s->home);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreate(&s->merged);
	if (e == hipSuccess) e = hipEventCreate(&s->merge_start);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->home_stream, hipStreamNonBlocking);
	if (e != hipSuccess)
	{
		hnsw_gpu_sharded_destroy(s);
		return fail(HNSW_GPU_ERR_HIP, "sharded index set-up failed: %s", hipGetErrorString(e));
	}
	*out = s;
	return HNSW_GPU_OK;
}

extern "C" size_t hnsw_gpu_sharded_nshards(const hnsw_gpu_sharded *s) { return s ? s->shards.size() : 0; }

static int grow(char **p, size_t *have, size_t want)
{
	if (want <= *have) return HNSW_GPU_OK;
	if (*p) (void) hipFree(*p);
	*p = nullptr; *have = 0;
	HIPCHK(hipMalloc((void **) p, want));
	*have = want;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_sharded_search_dev(hnsw_gpu_sharded *s, const coord_t *d_queries, size_t nq, size_t ef,
										   label_t *d_labels, dist_t *d_dists, uint32_t *d_counts, void *stream_)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "sharded index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!d_queries || !d_labels || !d_counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	std::lock_guard<std::mutex> lk(s->mu);
	hipStream_t stream = (hipStream_t) stream_;
	const size_t ns = s->shards.size(), dim = s->shards[0]->meta.dim;
	// one result block per shard on the home device: [labels nq*ef | dists nq*ef | counts nq]
	const size_t o_d = round_up(nq * ef * 8, 256), o_c = o_d + round_up(nq * ef * 4, 256), block = o_c + round_up(nq * 4, 256);
	HIPCHK(hipSetDevice(s->home));
	int rc = grow(&s->gather, &s->gather_bytes, ns * block);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s->ready, stream));
	for (size_t i = 0; i < ns; i++)
	{
		hnsw_gpu_index *ix = s->shards[i];
		HIPCHK(hipSetDevice(ix->device));
		HIPCHK(hipStreamWaitEvent(s->streams[i], s->ready, 0));
		// a call on ANOTHER user stream than the previous one: its shard kernels must not overwrite `gather` (and the
		// shards' result blocks) while the previous call's merge still reads it
		if (s->merged_set) HIPCHK(hipStreamWaitEvent(s->streams[i], s->merged, 0));
		const float *q = d_queries;
		if (ix->device != s->home)                  // the shard reads its queries from its own HBM
		{
			rc = grow((char **) &s->q_local[i], &s->q_cap[i], nq * dim * 4);
			if (rc) return rc;
			HIPCHK(hipMemcpyPeerAsync(s->q_local[i], ix->device, d_queries, s->home, nq * dim * 4, s->streams[i]));
			q = s->q_local[i];
		}
		char *blk = s->gather + i * block;
		if (!s->direct[i])
		{
			rc = grow(&s->out_local[i], &s->out_cap[i], block);
			if (rc) return rc;
			blk = s->out_local[i];
		}
		// per-shard searchKnn (hnswalg.cpp:234-252); with peer access the kernel's result stores land in the
		// home device's memory directly — no copy step, no collective
		rc = launch_search(ix, s->ws[i], q, dim, nq, ef, 0, (uint64_t *) blk, nullptr, (float *) (blk + o_d), (uint32_t *) (blk + o_c),
						   nullptr, s->streams[i]);
		if (rc) return rc;
		if (!s->direct[i])
			HIPCHK(hipMemcpyPeerAsync(s->gather + i * block, s->home, blk, ix->device, block, s->streams[i]));
		HIPCHK(hipEventRecord(s->done[i], s->streams[i]));
	}
	HIPCHK(hipSetDevice(s->home));
	for (size_t i = 0; i < ns; i++) HIPCHK(hipStreamWaitEvent(stream, s->done[i], 0));
	HIPCHK(hipEventRecord(s->merge_start, stream));
	rc = hnsw_gpu_merge_topk_strided_dev(s->home, (const label_t *) s->gather, block / 8, (const dist_t *) (s->gather + o_d), block / 4,
										 ns, nq, ef, d_labels, d_dists, d_counts, stream);
	if (rc) return rc;
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipEventRecord(s->merged, stream));
	s->merged_set = true;
	s->timed = true;
	return HNSW_GPU_OK;
}

// Where the time of the last hnsw_gpu_sharded_search[_dev] call went, per shard, from HIP events on each shard's own device:
// search_ms[i] = shard i's search kernel, peer_ms[i] = what followed it on that shard's stream until its results were in the home
// device's buffer (0 when the kernel stores them there itself through peer access: then the xGMI stores are part of search_ms;
// otherwise the staged peer copy), *merge_ms = the merge kernel on the home device.  Arrays of hnsw_gpu_sharded_nshards values;
// waits for the call to finish.
extern "C" int hnsw_gpu_sharded_last_ms(hnsw_gpu_sharded *s, float *search_ms, float *peer_ms, float *merge_ms, int *direct)
{
	if (!s || !search_ms || !peer_ms || !merge_ms) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(s->mu);
	if (!s->timed) return fail(HNSW_GPU_ERR_ARG, "no sharded search has run yet");
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipEventSynchronize(s->merged));
	HIPCHK(hipEventElapsedTime(merge_ms, s->merge_start, s->merged));
	for (size_t i = 0; i < s->shards.size(); i++)
	{
		SearchWs *w = s->ws[i];
		if (w->launches == 0) return fail(HNSW_GPU_ERR_INTERNAL, "shard %zu has no launch on record", i);
		const int evi = (int) ((w->launches - 1) % SearchWs::EV_RING);
		HIPCHK(hipSetDevice(s->shards[i]->device));
		HIPCHK(hipEventSynchronize(s->done[i]));
		HIPCHK(hipEventElapsedTime(&search_ms[i], w->ev0[evi], w->ev1[evi]));
		HIPCHK(hipEventElapsedTime(&peer_ms[i], w->ev1[evi], s->done[i]));
		if (direct) direct[i] = s->direct[i] ? 1 : 0;
	}
	HIPCHK(hipSetDevice(s->home));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_sharded_search(hnsw_gpu_sharded *s, const coord_t *queries, size_t nq, size_t ef,
									   label_t *labels, dist_t *dists, uint32_t *counts)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "sharded index is NULL");
	if (nq == 0) return HNSW_GPU_OK;
	if (!queries || !labels || !counts) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (ef == 0 || ef >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef %zu out of range", ef);
	const size_t dim = s->shards[0]->meta.dim;
	const size_t qb = round_up(nq * dim * 4, 256), lb = round_up(nq * ef * 8, 256), db = round_up(nq * ef * 4, 256),
				 cb = round_up(nq * 4, 256);
	char *p;
	{
		std::lock_guard<std::mutex> lk(s->mu);
		HIPCHK(hipSetDevice(s->home));
		int rc = grow(&s->io, &s->io_bytes, qb + lb + db + cb);
		if (rc) return rc;
		p = s->io;
	}
	float *dq = (float *) p; uint64_t *dl = (uint64_t *) (p + qb); float *dd = (float *) (p + qb + lb);
	uint32_t *dc = (uint32_t *) (p + qb + lb + db);
	HIPCHK(hipMemcpyAsync(dq, queries, nq * dim * 4, hipMemcpyHostToDevice, s->home_stream));
	int rc = hnsw_gpu_sharded_search_dev(s, dq, nq, ef, dl, dd, dc, s->home_stream);
	if (rc) return rc;
	HIPCHK(hipSetDevice(s->home));
	HIPCHK(hipMemcpyAsync(labels, dl, nq * ef * 8, hipMemcpyDeviceToHost, s->home_stream));
	if (dists) HIPCHK(hipMemcpyAsync(dists, dd, nq * ef * 4, hipMemcpyDeviceToHost, s->home_stream));
	HIPCHK(hipMemcpyAsync(counts, dc, nq * 4, hipMemcpyDeviceToHost, s->home_stream));
	HIPCHK(hipStreamSynchronize(s->home_stream));
	return HNSW_GPU_OK;
}
