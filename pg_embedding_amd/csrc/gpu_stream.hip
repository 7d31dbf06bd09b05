// gpu_stream.hip — streams: ONE resident search launch that the host feeds while it runs (csrc/device_search.h, "Stream mode")
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"

// ------------------------------------------------------------------------------------
// streams: ONE resident search launch that the host feeds while it runs (device_search.h, "Stream mode")
// ------------------------------------------------------------------------------------
struct hnsw_gpu_stream
{
	hnsw_gpu_ctx *ctx = nullptr;
	size_t ef = 0, ring = 0, dim = 0;
	char *pin = nullptr;                      // pinned, coherent: [queries | labels | dists | counts | flags | control words]
	float *Q = nullptr; label_t *L = nullptr; dist_t *D = nullptr; uint32_t *C = nullptr; uint32_t *F = nullptr;
	uint32_t *host_ctl = nullptr;             // [0] = queries published so far, [1] = stop
	uint32_t *dev_ctl = nullptr;              // the doorbell wave's device copies
	unsigned walkers = 0;
};

extern "C" int hnsw_gpu_stream_close(hnsw_gpu_stream *s);

extern "C" int hnsw_gpu_stream_open(hnsw_gpu_ctx *c, size_t ef, size_t ring, unsigned walkers, hnsw_gpu_stream **out)
{
	if (!c || !out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ring < 64 || ring > ((size_t) 1 << 20) || (ring & (ring - 1))) return fail(HNSW_GPU_ERR_ARG, "ring must be a power of two in [64, 2^20]");
	if (ef == 0 || ef > 512) return fail(HNSW_GPU_ERR_ARG, "a stream needs ef <= 512 (the team form of the beam kernel)");
	hnsw_gpu_index *ix = c->ix;
	HIPCHK(hipSetDevice(ix->device));
	if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	hnsw_gpu_stream *s = new (std::nothrow) hnsw_gpu_stream();
	if (!s) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	s->ctx = c; s->ef = ef; s->ring = ring; s->dim = ix->meta.dim; s->walkers = walkers ? walkers : 4u;
	const size_t qb = round_up(ring * s->dim * 4, 256), lb = round_up(ring * ef * 8, 256), db = round_up(ring * ef * 4, 256),
				 cb = round_up(ring * 4, 256), fb = round_up(ring * 4, 256);
	hipError_t e = hipHostMalloc((void **) &s->pin, qb + lb + db + cb + fb + 256, hipHostMallocCoherent);
	if (e == hipSuccess) e = hipMalloc((void **) &s->dev_ctl, STREAM_COPIES * STREAM_COPY_WORDS * 4);
	if (e == hipSuccess) e = hipMemset(s->dev_ctl, 0, STREAM_COPIES * STREAM_COPY_WORDS * 4);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		if (s->pin) (void) hipHostFree(s->pin);
		if (s->dev_ctl) (void) hipFree(s->dev_ctl);
		delete s;
		return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "stream buffers: %s", hipGetErrorString(e));
	}
	memset(s->pin, 0, qb + lb + db + cb + fb + 256);
	s->Q = (float *) s->pin; s->L = (label_t *) (s->pin + qb); s->D = (dist_t *) (s->pin + qb + lb);
	s->C = (uint32_t *) (s->pin + qb + lb + db); s->F = (uint32_t *) (s->pin + qb + lb + db + cb);
	s->host_ctl = (uint32_t *) (s->pin + qb + lb + db + cb + fb);
	int rc;
	{
		std::unique_lock<std::recursive_mutex> lock_(ix->mu);       // the "next launch only" fields and the launch are one step
		c->ws.done_next = s->F;
		c->ws.stream_host_next = s->host_ctl; c->ws.stream_dev_next = s->dev_ctl;
		c->ws.stream_ring_next = (uint32_t) ring; c->ws.stream_walkers_next = s->walkers;
#ifdef PGEMB_SIMT_EMULATOR
		simt::next_launch_is_resident();                         // (the CPU tier's emulator runs every other launch at the call)
#endif
		rc = launch_search(ix, &c->ws, s->Q, s->dim, ring, ef, 0, s->L, nullptr, s->D, s->C, nullptr, c->stream);
		c->ws.done_next = nullptr; c->ws.stream_host_next = nullptr; c->ws.stream_dev_next = nullptr;
	}
	if (rc)
	{
		(void) hipHostFree(s->pin); (void) hipFree(s->dev_ctl);
		delete s;
		return rc;
	}
	*out = s;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_stream_buffers(hnsw_gpu_stream *s, coord_t **queries, label_t **labels, dist_t **dists, uint32_t **counts, uint32_t **flags)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	if (queries) *queries = s->Q;
	if (labels) *labels = s->L;
	if (dists) *dists = s->D;
	if (counts) *counts = s->C;
	if (flags) *flags = s->F;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_stream_publish(hnsw_gpu_stream *s, uint32_t published_total)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	// Monotonic (a counter mod 2^32, compared by signed difference): several producer threads may publish, and a call that arrives
	// late with a smaller count must not take the word back.  Release: everything written into the slots before is visible before it.
	uint32_t cur = __atomic_load_n(&s->host_ctl[0], __ATOMIC_RELAXED);
	while ((int32_t) (published_total - cur) > 0 &&
		   !__atomic_compare_exchange_n(&s->host_ctl[0], &cur, published_total, true, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) {}
	return HNSW_GPU_OK;
}

// 1 = the stream's launch is still on the device, 0 = it has left (stopped, aborted or failed), < 0 = error
extern "C" int hnsw_gpu_stream_alive(hnsw_gpu_stream *s)
{
	if (!s) return fail(HNSW_GPU_ERR_ARG, "stream is NULL");
	const int idle = hnsw_gpu_ctx_idle(s->ctx);
	return idle < 0 ? idle : (idle ? 0 : 1);
}

static int stream_end(hnsw_gpu_stream *s, bool keep_buffers);
extern "C" int hnsw_gpu_stream_close(hnsw_gpu_stream *s) { return stream_end(s, false); }
/* the same stop, but the ring AND the handle stay allocated (leaked on purpose): for a host that could not prove that none of its
 * threads is still reading or writing the ring it was given (hnsw_gpu_stream_buffers) — or still about to call
 * hnsw_gpu_stream_publish on this handle, which reads s->host_ctl — when it had to give the stream up.  After this call the handle
 * stays VALID for _publish and _buffers (they touch only the leaked memory; nothing listens any more) and for nothing else. */
extern "C" int hnsw_gpu_stream_abandon(hnsw_gpu_stream *s) { return stream_end(s, true); }

static int stream_end(hnsw_gpu_stream *s, bool keep_buffers)
{
	if (!s) return HNSW_GPU_OK;
	hnsw_gpu_ctx *c = s->ctx;
	(void) hipSetDevice(c->ix->device);
	__atomic_store_n(&s->host_ctl[1], 1u, __ATOMIC_SEQ_CST);
	// every wave leaves at its next look (a walking wave after its query: under a millisecond); a launch that does not is a
	// hung launch: its workspace's abort word, then the wait again
	int rc = HNSW_GPU_OK;
	const int64_t t0 = now_ms();
	bool asked = false;
	while (hipStreamQuery(c->stream) == hipErrorNotReady)
	{
		if (!asked && now_ms() - t0 > 2000)
		{
			std::lock_guard<std::mutex> g(g_ws_mu);
			(void) abort_ws_locked(&c->ws);
			asked = true;
		}
		if (now_ms() - t0 > 1000ll * poll_limit_s()) { rc = fail(HNSW_GPU_ERR_INTERNAL, "the stream's launch did not end"); break; }
		std::this_thread::sleep_for(std::chrono::microseconds(20));
	}
	(void) hipGetLastError();
	if (rc == HNSW_GPU_OK && !keep_buffers)
	{
		(void) hipHostFree(s->pin);
		(void) hipFree(s->dev_ctl);
	}                                                            // (a launch that never ended may still write them: leaked on purpose)
	// An abandoned stream keeps its handle too: the caller abandons exactly because one of its threads may still be inside the ring, and
	// such a thread ends with hnsw_gpu_stream_publish(s, ...), which reads s->host_ctl and stores through it (freeing `s`
	// here was a use-after-free and possibly a wild write).  ~100 bytes per abandoned stream, an event the server logs.
	if (!keep_buffers) delete s;
	return rc;
}

// 1 = the context's last launch has left the device, 0 = still running, < 0 = error.
extern "C" int hnsw_gpu_ctx_idle(hnsw_gpu_ctx *c)
{
	if (!c) return fail(HNSW_GPU_ERR_ARG, "context is NULL");
	if (c->ws.launches == 0) return 1;
	const int evi = (int) ((c->ws.launches - 1) % SearchWs::EV_RING);
	hipError_t e = hipEventQuery(c->ws.ev1[evi]);
	if (e == hipSuccess) return 1;
	if (e == hipErrorNotReady) { (void) hipGetLastError(); return 0; }
	return fail(HNSW_GPU_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

