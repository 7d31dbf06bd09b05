// hnsw_gpu.hip — C-ABI implementation (include/hnsw_gpu.h) of the MI355X HNSW hot path: errors, configuration, search workspaces +
// watchdog, and the device mirror itself (create / import / export / append / reserve).  The other entry points live in the
// gpu_*.hip units beside this one (csrc/gpu_host.h lists them).  gfx950 only; plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}


extern "C" const char *hnsw_gpu_last_error(void) { return g_err; }

extern "C" int hnsw_gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}


// ------------------------------------------------------------------------------------
// configuration: resolved ONCE per process, never on a call path
// ------------------------------------------------------------------------------------
// Every knob of the library lives in one table of optional integers.  The table is filled from the environment at the first
// use (std::call_once) — the OPERATIONAL knobs only, the ones a deployment may want to set (include/hnsw_gpu.h,
// INTEGRATION.md "environment") — and after that a launch reads plain words: no getenv, no parsing, no locale on the path of
// a 0.4 ms call.  A host that wants another value later says so explicitly (hnsw_gpu_config_set; hnsw_gpu_config_reload re-reads
// the environment): that is how the test tiers flip kernel forms inside one process.  TEST knobs (forms and shapes the host code
// never picks by itself, forced so that every compiled kernel is exercised) can only be set through that call; experiment knobs
// of rejected variants exist only in -DHNSW_EXPERIMENT builds, where the whole table is read from the environment.
struct KnobDef { const char *name; bool env; };
static const KnobDef g_knob_def[K_COUNT] = {
	{ "HNSW_GPU_BEAM", true }, { "HNSW_GPU_FORCE_LDS_HEAPS", true }, { "HNSW_GPU_TEAM", true }, { "HNSW_GPU_TEAM_MAX_NQ", true },
	{ "HNSW_GPU_WIDE_EF_MIN", true }, { "HNSW_GPU_REF_ORDER", true }, { "HNSW_GPU_NO_POLL", true }, { "HNSW_GPU_POLL_LIMIT_S", true },
	{ "HNSW_GPU_INSERT_FUSED", true }, { "HNSW_GPU_BLOCKS_PER_CU", true }, { "HNSW_GPU_STREAM_LIGHT", true },
	{ "HNSW_GPU_BEAM16", false }, { "HNSW_GPU_NARROW5", false }, { "HNSW_GPU_LEAN", false }, { "HNSW_GPU_HASH_ENTRIES", false }, { "HNSW_GPU_LDS_SET_MIN_WAVES", false },
	{ "HNSW_GPU_TEAM_SPEC", false }, { "HNSW_GPU_TEAM_WPB", false }, { "HNSW_GPU_NARROW_WPB", false }, { "HNSW_GPU_ABORT_POLL_LOG2", false }, { "HNSW_GPU_MAX_BLOCKS", false }, { "HNSW_GPU_SHARDED_NO_PEER", false },
	{ "HNSW_GPU_BF_BIG_MIN_BLOCKS", false },
#ifdef HNSW_EXPERIMENT
	{ "HNSW_GPU_WIDE_WAVES", false }, { "HNSW_GPU_SHAPE_12X1", false }, { "HNSW_GPU_TEAM_MAINS", false }, { "HNSW_GPU_TEAM_COUNTERS", false },
#endif
};
KnobVal g_knob[K_COUNT];
static std::once_flag g_knob_once;

static void knob_store(int k, const char *text)
{
	if (text && *text) { g_knob[k].v.store(atoll(text), std::memory_order_relaxed); g_knob[k].set.store(true, std::memory_order_release); }
	else g_knob[k].set.store(false, std::memory_order_release);
}

static void knobs_from_env(bool all)
{
	for (int k = 0; k < K_COUNT; k++)
	{
#ifdef HNSW_EXPERIMENT
		(void) all;
		knob_store(k, getenv(g_knob_def[k].name));
#else
		if (g_knob_def[k].env || all) knob_store(k, g_knob_def[k].env ? getenv(g_knob_def[k].name) : nullptr);
#endif
	}
}

void knobs_init() { std::call_once(g_knob_once, [] { knobs_from_env(false); }); }

extern "C" int hnsw_gpu_config_set(const char *name, const char *value)
{
	if (!name) return fail(HNSW_GPU_ERR_ARG, "knob name is NULL");
	knobs_init();
	for (int k = 0; k < K_COUNT; k++)
		if (strcmp(name, g_knob_def[k].name) == 0) { knob_store(k, value); return HNSW_GPU_OK; }
	return fail(HNSW_GPU_ERR_ARG, "unknown knob %s", name);
}

extern "C" int hnsw_gpu_config_get(const char *name, long long *value)
{
	if (!name || !value) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	knobs_init();
	for (int k = 0; k < K_COUNT; k++)
		if (strcmp(name, g_knob_def[k].name) == 0)
		{
			if (!knob_is_set(k)) return 1;                  // known, at its default
			*value = knob(k, 0);
			return HNSW_GPU_OK;
		}
	return fail(HNSW_GPU_ERR_ARG, "unknown knob %s", name);
}

extern "C" void hnsw_gpu_config_reload(void)
{
	knobs_init();
	knobs_from_env(true);
}

// ------------------------------------------------------------------------------------
// the device mirror
// ------------------------------------------------------------------------------------
// Per-stream search state: the slots' visited bitmaps + logs, the ticket word and the HIP-event ring.
// Every mirror owns one (used by the plain entry points); hnsw_gpu_ctx adds more so that batches on
// different streams can be in flight at the same time.

// ------------------------------------------------------------------------------------
// Abort + watchdog.  No wait inside the kernels is unbounded, so a launch that never ends would be a
// bug nobody has thought of; the abort word makes such a launch cost its caller's patience instead of
// the device: every wave reads the abort word at the top of a query / every 256 hops and leaves.
// Every workspace is registered here so that ANY thread (a test watchdog, a signal-safe helper thread,
// the library's own watchdog: HNSW_GPU_WATCHDOG_S=<seconds>) can reach it without the mirror's lock —
// the thread that owns the lock is the one that is stuck.
// ------------------------------------------------------------------------------------
// (never destroyed: the watchdog thread is detached and may still be looking at them while the process exits)
std::mutex &g_ws_mu = *new std::mutex;
static std::vector<SearchWs *> &g_ws_all = *new std::vector<SearchWs *>;
static bool g_watchdog_started = false;

int64_t now_ms()
{
	return (int64_t) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// g_ws_mu held.  The abort word lives in pinned host memory: a plain store, no HIP call, nothing that could queue behind
// the launch it is meant to end.
int abort_ws_locked(SearchWs *w)
{
	if (!w->abort_host) return 0;
	__atomic_store_n(&w->abort_sent, 1, __ATOMIC_SEQ_CST);
	__atomic_add_fetch(&w->abort_requests, 1u, __ATOMIC_SEQ_CST);
	__atomic_store_n(w->abort_host, 1u, __ATOMIC_SEQ_CST);
	return 1;
}

extern "C" int hnsw_gpu_abort_all(void)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	int n = 0;
	for (SearchWs *w : g_ws_all) n += abort_ws_locked(w);
	return n;
}

static void watchdog_main(int limit_s)
{
	for (;;)
	{
		std::this_thread::sleep_for(std::chrono::milliseconds(500));
		std::lock_guard<std::mutex> g(g_ws_mu);
		const int64_t now = now_ms();
		for (SearchWs *w : g_ws_all)
		{
			const int64_t since = __atomic_load_n(&w->busy_since_ms, __ATOMIC_SEQ_CST);
			if (since == 0 || now - since < (int64_t) limit_s * 1000 || __atomic_load_n(&w->abort_sent, __ATOMIC_SEQ_CST)) continue;
			const uint64_t l = __atomic_load_n(&w->launches, __ATOMIC_SEQ_CST);
			if (l == 0) continue;
			hipEvent_t ev = w->ev1[(l - 1) % SearchWs::EV_RING];
			if (!ev || hipEventQuery(ev) != hipErrorNotReady)
			{
				// finished (or unknown): idle unless a newer launch has stamped it meanwhile
				int64_t expect = since;
				(void) __atomic_compare_exchange_n(&w->busy_since_ms, &expect, (int64_t) 0, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
				continue;
			}
			// The stamp was taken when the launch was ENQUEUED.  A deep queue of healthy asynchronous launches may keep it waiting
			// longer than the limit: while its start event has not completed it is not running, so the clock restarts now.
			hipEvent_t ev_start = w->ev0[(l - 1) % SearchWs::EV_RING];
			if (ev_start && hipEventQuery(ev_start) == hipErrorNotReady)
			{
				int64_t expect = since;
				(void) __atomic_compare_exchange_n(&w->busy_since_ms, &expect, now, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
				continue;
			}
			fprintf(stderr, "hnsw_gpu watchdog: search kernel %s on device %d has been running for %lld s (limit %d s): aborting it\n",
					w->kname, w->device, (long long) ((now - since) / 1000), limit_s);
			(void) abort_ws_locked(w);
		}
	}
}

// g_ws_mu NOT held
static void ws_register(SearchWs *w)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	g_ws_all.push_back(w);
	if (!g_watchdog_started)
	{
		g_watchdog_started = true;
		const char *e = getenv("HNSW_GPU_WATCHDOG_S");
		if (e && atoi(e) > 0) std::thread(watchdog_main, atoi(e)).detach();
	}
}

static void ws_unregister(SearchWs *w)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	g_ws_all.erase(std::remove(g_ws_all.begin(), g_ws_all.end(), w), g_ws_all.end());
}

int ws_init(SearchWs *w)
{
	HIPCHK(hipMalloc(&w->ticket, 64));
	HIPCHK(hipMemset(w->ticket, 0, 64));
	HIPCHK(hipMalloc(&w->health, HEALTH_WORDS * 4));
	HIPCHK(hipMemset(w->health, 0, HEALTH_WORDS * 4));
	HIPCHK(hipHostMalloc((void **) &w->abort_host, 64, hipHostMallocDefault));
	memset(w->abort_host, 0, 64);
	(void) hipGetDevice(&w->device);
	for (int i = 0; i < SearchWs::EV_RING; i++)
	{
		HIPCHK(hipEventCreate(&w->ev0[i]));
		HIPCHK(hipEventCreate(&w->ev1[i]));
	}
	ws_register(w);
	return HNSW_GPU_OK;
}

void ws_free(SearchWs *w)
{
	ws_unregister(w);
	if (w->health) (void) hipFree(w->health);
	if (w->abort_host) (void) hipHostFree(w->abort_host);
	if (w->vis) (void) hipFree(w->vis);
	if (w->beam) (void) hipFree(w->beam);
	if (w->sets) (void) hipFree(w->sets);
	if (w->vlog) (void) hipFree(w->vlog);
	if (w->team_dbg) (void) hipFree(w->team_dbg);
	if (w->ticket) (void) hipFree(w->ticket);
	for (int i = 0; i < SearchWs::EV_RING; i++)
	{
		if (w->ev0[i]) (void) hipEventDestroy(w->ev0[i]);
		if (w->ev1[i]) (void) hipEventDestroy(w->ev1[i]);
	}
	*w = SearchWs();
}


int ensure_scratch(hnsw_gpu_index *ix, size_t bytes)
{
	if (bytes <= ix->scratch_bytes) return HNSW_GPU_OK;
	if (ix->scratch) (void) hipFree(ix->scratch);
	ix->scratch = nullptr; ix->scratch_bytes = 0;
	HIPCHK(hipMalloc(&ix->scratch, bytes));
	ix->scratch_bytes = bytes;
	return HNSW_GPU_OK;
}

static int check_meta(const HnswMetadata *m)
{
	if (!m) return fail(HNSW_GPU_ERR_ARG, "meta is NULL");
	if (m->dim == 0 || m->dim > (1u << 20)) return fail(HNSW_GPU_ERR_ARG, "unsupported dim %zu", m->dim);
	if (m->maxM == 0 || m->maxM > 4096) return fail(HNSW_GPU_ERR_ARG, "unsupported maxM %zu", m->maxM);
	if ((int) m->dist_func < 0 || (int) m->dist_func > 2) return fail(HNSW_GPU_ERR_ARG, "bad dist_func %d", (int) m->dist_func);
	// element image offsets must be the ones of embedding.c:225-228
	if (m->offset_data != (m->maxM + 1) * sizeof(idx_t) || m->offset_label != m->offset_data + m->dim * sizeof(coord_t) ||
		m->size_data_per_element != m->offset_label + sizeof(label_t))
		return fail(HNSW_GPU_ERR_ARG, "meta offsets do not describe [count|links|vector|label]");
	return HNSW_GPU_OK;
}

// The mirror's three arrays come from ONE allocation, each starting on a 2 MiB boundary of it (rows | links | labels).  A walk is a
// chain of dependent random reads — a 128-byte link list, then a handful of rows — so what it pays per read is latency, and a part of
// that is address translation: three separate allocations land wherever the process's earlier allocations left holes, and round 5
// saw the narrow-row launch 40 % slower on every launch of some processes with nothing but the allocation history different
// (profiles/r5af_*).  One block whose placement inside itself is fixed takes the history out: the driver maps a large aligned
// allocation with its largest fragments, and the three arrays keep the same relative position in every process.
static const size_t MIRROR_ALIGN = (size_t) 2 << 20;
static hipError_t alloc_mirror(const hnsw_gpu_index *ix, size_t cap, char **arena, size_t *arena_bytes, float **vec, uint32_t **links, uint64_t **labels)
{
	const size_t vb = round_up(cap * ix->stride * sizeof(float), MIRROR_ALIGN), lb = round_up(cap * ix->lstride * sizeof(uint32_t), MIRROR_ALIGN),
				 bb = round_up(cap * sizeof(uint64_t), MIRROR_ALIGN);
	char *a = nullptr;
	const hipError_t e = hipMalloc(&a, vb + lb + bb + MIRROR_ALIGN);       // (+ one granule: the arrays are aligned inside the block whatever its base)
	if (e != hipSuccess) return e;
	char *base = (char *) round_up((size_t) a, MIRROR_ALIGN);
	*arena = a; *arena_bytes = vb + lb + bb + MIRROR_ALIGN;
	*vec = (float *) base; *links = (uint32_t *) (base + vb); *labels = (uint64_t *) (base + vb + lb);
	return hipSuccess;
}

static int alloc_index(const HnswMetadata *meta, size_t capacity, int device, hnsw_gpu_index **out)
{
	int rc = check_meta(meta);
	if (rc) return rc;
	int ndev = hnsw_gpu_device_count();
	if (ndev <= 0) return fail(HNSW_GPU_ERR_NODEVICE, "no HIP device visible (this library has no CPU path)");
	if (device < 0 || device >= ndev) return fail(HNSW_GPU_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
	if (capacity >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "capacity exceeds idx_t range");
	HIPCHK(hipSetDevice(device));
	hnsw_gpu_index *ix = new (std::nothrow) hnsw_gpu_index();
	if (!ix) return fail(HNSW_GPU_ERR_NOMEM, "out of host memory");
	ix->meta = *meta;
	ix->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess)
	{
		ix->num_cu = prop.multiProcessorCount;
		ix->gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
	}
	if (ix->num_cu <= 0) ix->num_cu = 256;
	{
		int maxlds = 0;
		if (hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && maxlds > 0) ix->max_lds = (size_t) maxlds;
	}
	ix->cap = capacity ? capacity : 1;
	ix->stride = (uint32_t) round_up(meta->dim, 4);
	ix->lstride = (uint32_t) round_up(meta->maxM, 16);
	hipError_t e = alloc_mirror(ix, ix->cap, &ix->arena, &ix->arena_bytes, &ix->vec, &ix->links, &ix->labels);
	if (e == hipSuccess) e = hipMalloc(&ix->misc, 64);
	if (e != hipSuccess)
	{
		hnsw_gpu_index_destroy(ix);
		return fail(e == hipErrorOutOfMemory ? HNSW_GPU_ERR_NOMEM : HNSW_GPU_ERR_HIP, "index allocation failed: %s",
					hipGetErrorString(e));
	}
	(void) hipMemset(ix->misc, 0, 64);
	{
		int rc2 = ws_init(&ix->ws);
		if (rc2) { hnsw_gpu_index_destroy(ix); return rc2; }
	}
	*out = ix;
	return HNSW_GPU_OK;
}

extern "C" void hnsw_gpu_index_destroy(hnsw_gpu_index *ix)
{
	if (!ix) return;
	(void) hipSetDevice(ix->device);
	if (ix->arena) (void) hipFree(ix->arena);
	ws_free(&ix->ws);
	if (ix->misc) (void) hipFree(ix->misc);
	if (ix->scratch) (void) hipFree(ix->scratch);
	if (ix->pin) (void) hipHostFree(ix->pin);
	if (ix->bld) (void) hipFree(ix->bld);
	if (ix->ins) (void) hipFree(ix->ins);
	if (ix->xnorm) (void) hipFree(ix->xnorm);
	if (ix->bf) (void) hipFree(ix->bf);
	if (ix->bf_e0) (void) hipEventDestroy(ix->bf_e0);
	if (ix->bf_e1) (void) hipEventDestroy(ix->bf_e1);
	if (ix->hb0) (void) hipEventDestroy(ix->hb0);
	if (ix->hb1) (void) hipEventDestroy(ix->hb1);
	delete ix;
}

extern "C" size_t hnsw_gpu_index_count(const hnsw_gpu_index *ix) { return ix ? ix->n : 0; }
extern "C" size_t hnsw_gpu_index_capacity(const hnsw_gpu_index *ix) { return ix ? ix->cap : 0; }
extern "C" int    hnsw_gpu_index_device(const hnsw_gpu_index *ix) { return ix ? ix->device : -1; }

// ------------------------------------------------------------------------------------
// element image <-> mirror
// ------------------------------------------------------------------------------------

// One wavefront per element: [count|links|vector|label] -> links row / padded vector row / label.
// Everything in the image is 4-byte aligned only (embedding.c:226), so it is read as u32 words.
__global__ __launch_bounds__(256) void import_elements_kernel(const uint32_t *__restrict__ raw, size_t elem_words,
															  uint32_t first, uint32_t count, uint32_t n_total,
															  uint32_t dim, uint32_t stride, uint32_t maxM, uint32_t lstride,
															  float *vec, uint32_t *links, uint64_t *labels, uint32_t *bad)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	const uint32_t *src = raw + (size_t) w * elem_words;
	const size_t e = (size_t) first + w;
	uint32_t cnt = src[0];
	if (cnt > maxM) { if (lane == 0) atomicAdd(bad, 1u); cnt = maxM; }
	for (uint32_t j = lane; j < lstride; j += 64)
	{
		uint32_t t = LINK_NONE;
		if (j < cnt)
		{
			t = src[1 + j];
			if (t >= n_total) { atomicAdd(bad, 1u); t = LINK_NONE; }
			for (uint32_t k = 0; k < j && t != LINK_NONE; k++)      // keep the first occurrence only
				if (src[1 + k] == t) t = LINK_NONE;
		}
		links[e * lstride + j] = t;
	}
	const uint32_t *v = src + (maxM + 1);
	for (uint32_t c = lane; c < stride; c += 64)
		vec[e * stride + c] = (c < dim) ? __uint_as_float(v[c]) : 0.f;
	if (lane == 0)
	{
		const uint32_t *l = v + dim;
		labels[e] = (uint64_t) l[0] | ((uint64_t) l[1] << 32);
	}
}

__global__ __launch_bounds__(256) void export_elements_kernel(uint32_t *__restrict__ raw, size_t elem_words,
															  uint32_t first, uint32_t count,
															  uint32_t dim, uint32_t stride, uint32_t maxM, uint32_t lstride,
															  const float *vec, const uint32_t *links, const uint64_t *labels)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	uint32_t *dst = raw + (size_t) w * elem_words;
	const size_t e = (size_t) first + w;
	// links are stored compacted (no holes) with their count in front
	uint32_t cnt = 0;
	for (uint32_t j0 = 0; j0 < maxM; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		uint32_t t = (j < lstride && j < maxM) ? links[e * lstride + j] : LINK_NONE;
		const uint64_t m = __ballot(t != LINK_NONE);
		if (t != LINK_NONE) dst[1 + cnt + lane_rank(m)] = t;
		cnt += (uint32_t) __builtin_popcountll(m);
	}
	for (uint32_t j = cnt + lane; j < maxM; j += 64) dst[1 + j] = 0u;
	if (lane == 0) dst[0] = cnt;
	uint32_t *v = dst + (maxM + 1);
	for (uint32_t c = lane; c < dim; c += 64) v[c] = __float_as_uint(vec[e * stride + c]);
	if (lane == 0)
	{
		const uint64_t l = labels[e];
		v[dim] = (uint32_t) l;
		v[dim + 1] = (uint32_t) (l >> 32);
	}
}

// rows given as dim-strided floats (host order) -> padded rows, links cleared, labels set.
__global__ __launch_bounds__(256) void append_rows_kernel(const float *__restrict__ src, const uint64_t *__restrict__ src_labels,
														  uint32_t first, uint32_t count, uint32_t dim, uint32_t stride,
														  uint32_t lstride, float *vec, uint32_t *links, uint64_t *labels)
{
	const int lane = threadIdx.x & 63;
	const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= count) return;
	const size_t e = (size_t) first + w;
	for (uint32_t c = lane; c < stride; c += 64)
		vec[e * stride + c] = (c < dim) ? src[(size_t) w * dim + c] : 0.f;
	for (uint32_t j = lane; j < lstride; j += 64) links[e * lstride + j] = LINK_NONE;
	if (lane == 0) labels[e] = src_labels ? src_labels[w] : (uint64_t) e;
}

static const size_t STAGE_BYTES = (size_t) 128 << 20;   // host<->device staging granule (two of them for uploads)

// Re-import `count` host element images into element numbers [first, first+count); the mirror
// grows to cover them.  n_total bounds the link targets that are accepted.  Two staging buffers, a copy
// stream and a kernel stream: the copy of granule i+1 runs while granule i is scattered into the mirror, and
// nothing waits for the whole device (other mirrors' searches keep running).
int import_range(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count, size_t n_total)
{
	const HnswMetadata *meta = &ix->meta;
	const size_t esz = meta->size_data_per_element;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / esz);
	uint32_t *bad = ix->misc;
	if (count == 0) return HNSW_GPU_OK;
	HIPCHK(hipMemset(bad, 0, 4));
	const size_t nbuf = count > per ? 2 : 1;
	const size_t buf_bytes = round_up(std::min(per, count) * esz, 256);
	int rc0 = ensure_scratch(ix, nbuf * buf_bytes);                // staging lives in the mirror's scratch
	if (rc0) return rc0;
	hipStream_t s_copy = nullptr, s_kern = nullptr;
	hipEvent_t copied[2] = { nullptr, nullptr }, used[2] = { nullptr, nullptr };
	hipError_t e = hipSuccess;
	if (nbuf == 1)
	{
		// one granule (incremental updates, small mirrors): no pipeline to build — copy, scatter, and the read-back
		// of the error counter below waits for the kernel on the default stream
		e = hipMemcpy(ix->scratch, elements, count * esz, hipMemcpyHostToDevice);
		if (e == hipSuccess)
		{
			hipLaunchKernelGGL(import_elements_kernel, dim3((uint32_t) ((count + 3) / 4)), dim3(256), 0, 0, (const uint32_t *) ix->scratch,
							   esz / 4, (uint32_t) first, (uint32_t) count, (uint32_t) n_total, (uint32_t) meta->dim, ix->stride,
							   (uint32_t) meta->maxM, ix->lstride, ix->vec, ix->links, ix->labels, bad);
			e = hipGetLastError();
		}
		uint32_t nb1 = 0;
		if (e == hipSuccess) e = hipMemcpy(&nb1, bad, 4, hipMemcpyDeviceToHost);
		if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index upload failed: %s", hipGetErrorString(e));
		if (nb1) return fail(HNSW_GPU_ERR_ARG, "element image is corrupt: %u bad link counts / link targets", nb1);
		return HNSW_GPU_OK;
	}
	e = hipStreamCreate(&s_copy);
	if (e == hipSuccess) e = hipStreamCreate(&s_kern);
	for (int b = 0; b < 2 && e == hipSuccess; b++)
	{
		e = hipEventCreateWithFlags(&copied[b], hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&used[b], hipEventDisableTiming);
	}
	size_t chunk = 0;
	for (size_t off = 0; off < count && e == hipSuccess; off += per, chunk++)
	{
		const size_t cnt = std::min(per, count - off);
		const int b = (int) (chunk % nbuf);
		uint32_t *stage = (uint32_t *) ((char *) ix->scratch + (size_t) b * buf_bytes);
		if (chunk >= nbuf) e = hipStreamWaitEvent(s_copy, used[b], 0);       // the kernel that read this buffer is done
		if (e == hipSuccess) e = hipMemcpyAsync(stage, (const char *) elements + off * esz, cnt * esz, hipMemcpyHostToDevice, s_copy);
		if (e == hipSuccess) e = hipEventRecord(copied[b], s_copy);
		if (e == hipSuccess) e = hipStreamWaitEvent(s_kern, copied[b], 0);
		if (e != hipSuccess) break;
		const uint32_t blocks = (uint32_t) ((cnt + 3) / 4);
		hipLaunchKernelGGL(import_elements_kernel, dim3(blocks), dim3(256), 0, s_kern, stage, esz / 4, (uint32_t) (first + off),
						   (uint32_t) cnt, (uint32_t) n_total, (uint32_t) meta->dim, ix->stride, (uint32_t) meta->maxM,
						   ix->lstride, ix->vec, ix->links, ix->labels, bad);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipEventRecord(used[b], s_kern);
	}
	if (s_copy) { const hipError_t e2 = hipStreamSynchronize(s_copy); if (e == hipSuccess) e = e2; }
	if (s_kern) { const hipError_t e2 = hipStreamSynchronize(s_kern); if (e == hipSuccess) e = e2; }
	uint32_t nbad = 0;
	if (e == hipSuccess) e = hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
	for (int b = 0; b < 2; b++)
	{
		if (copied[b]) (void) hipEventDestroy(copied[b]);
		if (used[b]) (void) hipEventDestroy(used[b]);
	}
	if (s_copy) (void) hipStreamDestroy(s_copy);
	if (s_kern) (void) hipStreamDestroy(s_kern);
	if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index upload failed: %s", hipGetErrorString(e));
	if (nbad) return fail(HNSW_GPU_ERR_ARG, "element image is corrupt: %u bad link counts / link targets", nbad);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_create_from_flat(const HnswMetadata *meta, const void *elements, size_t n,
											   int device, hnsw_gpu_index **out)
{
	if (!out) return fail(HNSW_GPU_ERR_ARG, "out is NULL");
	if (n && !elements) return fail(HNSW_GPU_ERR_ARG, "elements is NULL");
	hnsw_gpu_index *ix = nullptr;
	int rc = alloc_index(meta, n, device, &ix);
	if (rc) return rc;
	rc = import_range(ix, elements, 0, n, n);
	if (rc) { hnsw_gpu_index_destroy(ix); return rc; }
	ix->n = n;
	*out = ix;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_create_empty(const HnswMetadata *meta, size_t capacity, int device, hnsw_gpu_index **out)
{
	if (!out) return fail(HNSW_GPU_ERR_ARG, "out is NULL");
	return alloc_index(meta, capacity, device, out);
}

extern "C" int hnsw_gpu_index_append_dev(hnsw_gpu_index *ix, const coord_t *d_vectors, const label_t *d_labels,
										 size_t n, void *stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (n == 0) return HNSW_GPU_OK;
	if (!d_vectors) return fail(HNSW_GPU_ERR_ARG, "vectors is NULL");
	if (ix->n + n > ix->cap) return fail(HNSW_GPU_ERR_ARG, "append exceeds capacity (%zu + %zu > %zu)", ix->n, n, ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	const uint32_t blocks = (uint32_t) ((n + 3) / 4);
	hipLaunchKernelGGL(append_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, d_vectors, d_labels,
					   (uint32_t) ix->n, (uint32_t) n, (uint32_t) ix->meta.dim, ix->stride, ix->lstride, ix->vec,
					   ix->links, ix->labels);
	HIPCHK(hipGetLastError());
	ix->n += n;
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_append(hnsw_gpu_index *ix, const coord_t *vectors, const label_t *labels, size_t n)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (n == 0) return HNSW_GPU_OK;
	if (!vectors) return fail(HNSW_GPU_ERR_ARG, "vectors is NULL");
	if (ix->n + n > ix->cap) return fail(HNSW_GPU_ERR_ARG, "append exceeds capacity (%zu + %zu > %zu)", ix->n, n, ix->cap);
	HIPCHK(hipSetDevice(ix->device));
	const size_t dim = ix->meta.dim;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / (dim * 4 + 8));
	int rc = ensure_scratch(ix, std::min(per, n) * (dim * 4 + 8) + 8);      // + alignment slack for the label array
	if (rc) return rc;
	for (size_t first = 0; first < n; first += per)
	{
		const size_t cnt = std::min(per, n - first);
		float *dv = (float *) ix->scratch;
		uint64_t *dl = (uint64_t *) ((char *) ix->scratch + round_up(cnt * dim * 4, 8));
		HIPCHK(hipMemcpy(dv, vectors + first * dim, cnt * dim * 4, hipMemcpyHostToDevice));
		if (labels) HIPCHK(hipMemcpy(dl, labels + first, cnt * 8, hipMemcpyHostToDevice));
		rc = hnsw_gpu_index_append_dev(ix, dv, labels ? dl : nullptr, cnt, nullptr);
		if (rc) return rc;
		HIPCHK(hipDeviceSynchronize());
	}
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_export_flat(hnsw_gpu_index *ix, void *elements)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !elements) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	HIPCHK(hipSetDevice(ix->device));
	const size_t esz = ix->meta.size_data_per_element;
	const size_t per = std::max<size_t>(1, STAGE_BYTES / esz);
	uint32_t *stage = nullptr;
	HIPCHK(hipMalloc(&stage, std::min(per, std::max<size_t>(ix->n, 1)) * esz));
	hipError_t e = hipSuccess;
	for (size_t first = 0; first < ix->n && e == hipSuccess; first += per)
	{
		const size_t cnt = std::min(per, ix->n - first);
		const uint32_t blocks = (uint32_t) ((cnt + 3) / 4);
		hipLaunchKernelGGL(export_elements_kernel, dim3(blocks), dim3(256), 0, 0, stage, esz / 4, (uint32_t) first,
						   (uint32_t) cnt, (uint32_t) ix->meta.dim, ix->stride, (uint32_t) ix->meta.maxM, ix->lstride,
						   ix->vec, ix->links, ix->labels);
		e = hipMemcpy((char *) elements + first * esz, stage, cnt * esz, hipMemcpyDeviceToHost);
	}
	(void) hipFree(stage);
	if (e != hipSuccess) return fail(HNSW_GPU_ERR_HIP, "index download failed: %s", hipGetErrorString(e));
	return HNSW_GPU_OK;
}

// vacuum flags of a batch of elements: one upload of the element numbers + one launch (a VACUUM flags many rows,
// embedding.c:883-946; doing them one blocking copy pair at a time is a synchronisation storm)
__global__ __launch_bounds__(256) void set_deleted_kernel(uint64_t *__restrict__ labels, const uint32_t *__restrict__ idx,
														 size_t count, int deleted)
{
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	const unsigned long long bit = 1ull << HNSW_LABEL_DELETED_BIT;
	unsigned long long *l = reinterpret_cast<unsigned long long *>(labels + idx[i]);
	if (deleted) atomicOr(l, bit); else atomicAnd(l, ~bit);        // atomic: the same element may be listed twice
}

extern "C" int hnsw_gpu_index_set_deleted_batch(hnsw_gpu_index *ix, const idx_t *idx, size_t count, int deleted)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (count == 0) return HNSW_GPU_OK;
	if (!idx) return fail(HNSW_GPU_ERR_ARG, "NULL element list");
	for (size_t i = 0; i < count; i++)
		if (idx[i] >= ix->n) return fail(HNSW_GPU_ERR_ARG, "bad element %u", (unsigned) idx[i]);
	HIPCHK(hipSetDevice(ix->device));
	int rc = ensure_scratch(ix, count * sizeof(uint32_t));
	if (rc) return rc;
	HIPCHK(hipMemcpy(ix->scratch, idx, count * sizeof(uint32_t), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(set_deleted_kernel, dim3((uint32_t) ((count + 255) / 256)), dim3(256), 0, 0,
					   ix->labels, (const uint32_t *) ix->scratch, count, deleted);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(0));
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_index_set_deleted(hnsw_gpu_index *ix, idx_t idx, int deleted)
{
	return hnsw_gpu_index_set_deleted_batch(ix, &idx, 1, deleted);
}

extern "C" int hnsw_gpu_index_reserve(hnsw_gpu_index *ix, size_t capacity)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "index is NULL");
	if (capacity <= ix->cap) return HNSW_GPU_OK;
	if (capacity >= 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "capacity exceeds idx_t range");
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipDeviceSynchronize());
	char *na = nullptr; size_t nab = 0;
	float *nv = nullptr; uint32_t *nl = nullptr; uint64_t *nb = nullptr;
	const hipError_t e = alloc_mirror(ix, capacity, &na, &nab, &nv, &nl, &nb);
	if (e != hipSuccess)
		return fail(HNSW_GPU_ERR_NOMEM, "cannot grow the mirror to %zu elements: %s", capacity, hipGetErrorString(e));
	HIPCHK(hipMemcpy(nv, ix->vec, ix->n * ix->stride * sizeof(float), hipMemcpyDeviceToDevice));
	HIPCHK(hipMemcpy(nl, ix->links, ix->n * ix->lstride * sizeof(uint32_t), hipMemcpyDeviceToDevice));
	HIPCHK(hipMemcpy(nb, ix->labels, ix->n * sizeof(uint64_t), hipMemcpyDeviceToDevice));
	(void) hipFree(ix->arena);
	ix->arena = na; ix->arena_bytes = nab;
	ix->vec = nv; ix->links = nl; ix->labels = nb;
	ix->cap = capacity;
	// the visited bitmaps are sized by capacity: drop them, the next search re-creates them
	if (ix->ws.vis) (void) hipFree(ix->ws.vis);
	if (ix->ws.vlog) (void) hipFree(ix->ws.vlog);
	ix->ws.vis = nullptr; ix->ws.vlog = nullptr; ix->ws.vis_slots = 0; ix->ws.vis_words = 0;
	ix->generation++;         // contexts notice and rebuild their bitmaps
	return HNSW_GPU_OK;
}

// Refresh part of the mirror from the host: element images of [first, first+count) replace what
// the mirror holds (links, vector, label); elements past the current end are added.  This is the
// incremental counterpart of create_from_flat for a host that tracks which pages changed
// (new elements, re-linked neighbours, vacuum flags) — SURVEY.md §8(f) rank 2.
extern "C" int hnsw_gpu_index_update_from_flat(hnsw_gpu_index *ix, const void *elements, size_t first, size_t count)
{
	if (!ix || (count && !elements)) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::unique_lock<std::recursive_mutex> lock_(ix->mu);
	if (first > ix->n) return fail(HNSW_GPU_ERR_ARG, "update would leave a gap (first %zu > count %zu)", first, ix->n);
	HIPCHK(hipSetDevice(ix->device));
	const size_t end = first + count;
	if (end > ix->cap)
	{
		int rc = hnsw_gpu_index_reserve(ix, end + end / 2);
		if (rc) return rc;
	}
	const size_t n_total = std::max(ix->n, end);
	int rc = import_range(ix, elements, first, count, n_total);
	if (rc) return rc;
	ix->n = n_total;
	ix->xnorm_n = 0;            // cached row norms are stale
	return HNSW_GPU_OK;
}

// Pinned host memory for the host-pointer entry points (NULL when there is no device / no memory).
extern "C" void *hnsw_gpu_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
	{
		(void) hipGetLastError();
		fail(HNSW_GPU_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

extern "C" void hnsw_gpu_host_free(void *p)
{
	if (p) (void) hipHostFree(p);
}

