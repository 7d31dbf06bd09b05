// gpu_scan.hip — batched distances (hnsw_dist_func over many rows) and exhaustive k-NN: canonical scan, MFMA filter (csrc/device_bf_mfma.h)
// One translation unit of libhnsw_gpu.so (csrc/gpu_host.h lists them); gfx950 only, plain HIP runtime, no framework types in any signature.
#include "gpu_host.h"
#include "device_bf_mfma.h"

// ------------------------------------------------------------------------------------
// batched distances (hnsw_dist_func over many rows)
// ------------------------------------------------------------------------------------
template <int FUNC>
__global__ __launch_bounds__(256) void dist_batch_kernel(const float *__restrict__ q, const float *__restrict__ rows,
														 uint32_t nrows, uint32_t dim, uint32_t stride, uint32_t nchunks,
														 uint32_t kiters, uint32_t qpad_floats, float *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float *qf = reinterpret_cast<float *>(smem);
	const float4 *q4 = reinterpret_cast<const float4 *>(smem);
	float *sums = reinterpret_cast<float *>(smem + (size_t) qpad_floats * 4) + (threadIdx.x >> 6) * 128;   // per wave
	for (uint32_t e = threadIdx.x; e < qpad_floats; e += blockDim.x)
	{
		const float t = q[e < dim ? e : dim - 1];
		qf[e] = (e < dim) ? t : 0.f;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, nchunks, kiters, lane);
	for (uint32_t base = wave * 64; base < nrows; base += nwaves * 64)
	{
		const uint32_t cnt = min(64u, nrows - base);
		auto direct = [base](uint32_t r) { return base + r; };
		score_rows<FUNC, 4, 2>(rows, stride, q4, nchunks, kiters, direct, cnt, sums, lane);
		wave_sync();
		const float d = finish_dist<FUNC>(sums[lane], sums[OUT2 + lane], qnorm);
		if ((uint32_t) lane < cnt) out[base + lane] = d;
		wave_sync();
	}
}

extern "C" int hnsw_gpu_dist_batch_dev(dist_func_t func, const coord_t *d_q, const coord_t *d_rows, size_t nrows,
									   size_t dim, size_t row_stride, dist_t *d_out, void *stream)
{
	if (nrows == 0) return HNSW_GPU_OK;
	if (!d_q || !d_rows || !d_out) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if ((int) func < 0 || (int) func > 2) return fail(HNSW_GPU_ERR_ARG, "bad dist_func %d", (int) func);
	if (dim == 0 || row_stride < dim || (row_stride & 3) || (((uintptr_t) d_rows) & 15))
		return fail(HNSW_GPU_ERR_ARG, "rows must be 16-byte aligned with stride %% 4 == 0 and stride >= dim");
	if (nrows >= 0xFFFFFFF0ull) return fail(HNSW_GPU_ERR_ARG, "too many rows");
	const uint32_t nchunks = (uint32_t) (row_stride / 4), kiters = (nchunks + 15) / 16;
	const uint32_t qpad = (uint32_t) round_up(kiters, 4) * 64;
	const size_t lds = (size_t) qpad * 4 + 4 * 128 * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "dim %zu too large", dim);
	const uint32_t blocks = (uint32_t) std::min<size_t>((nrows + 255) / 256, 256 * 8);
	hipStream_t s = (hipStream_t) stream;
	switch ((int) func)
	{
		case F_L2:
			hipLaunchKernelGGL(dist_batch_kernel<F_L2>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
		case F_COSINE:
			hipLaunchKernelGGL(dist_batch_kernel<F_COSINE>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
		default:
			hipLaunchKernelGGL(dist_batch_kernel<F_MANHATTAN>, dim3(blocks), dim3(256), lds, s, d_q, d_rows, (uint32_t) nrows,
							   (uint32_t) dim, (uint32_t) row_stride, nchunks, kiters, qpad, d_out);
			break;
	}
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_dist_batch(dist_func_t func, const coord_t *q, const coord_t *rows, size_t nrows, size_t dim,
								   dist_t *out)
{
	if (nrows == 0) return HNSW_GPU_OK;
	if (!q || !rows || !out) return fail(HNSW_GPU_ERR_ARG, "NULL buffer");
	if (dim == 0) return fail(HNSW_GPU_ERR_ARG, "dim is 0");
	if (hnsw_gpu_device_count() <= 0) return fail(HNSW_GPU_ERR_NODEVICE, "no HIP device visible (this library has no CPU path)");
	const size_t stride = round_up(dim, 4);
	// Small calls — the SQL operators hand over ONE pair per call (embedding.c:1037) — go through a
	// per-thread pinned staging area that the kernel reads and writes directly: no allocation, no copy
	// engine, one launch + one stream wait.
	const size_t small_bytes = (1 + nrows) * stride * 4 + round_up(nrows * 4, 16);
	if (small_bytes <= ((size_t) 256 << 10))
	{
		static thread_local char *pin = nullptr;
		static thread_local size_t pin_bytes = 0;
		static thread_local hipStream_t pin_stream = nullptr;
		static thread_local int pin_device = -1;
		int dev = 0;
		HIPCHK(hipGetDevice(&dev));
		if (pin_device != dev || pin_bytes < small_bytes)
		{
			if (pin) (void) hipHostFree(pin);
			if (pin_stream) (void) hipStreamDestroy(pin_stream);
			pin = nullptr; pin_bytes = 0; pin_stream = nullptr; pin_device = -1;
			HIPCHK(hipHostMalloc((void **) &pin, (size_t) 256 << 10, hipHostMallocDefault));
			HIPCHK(hipStreamCreateWithFlags(&pin_stream, hipStreamNonBlocking));
			pin_bytes = (size_t) 256 << 10;
			pin_device = dev;
		}
		float *hq = (float *) pin, *hr = hq + stride, *ho = (float *) (pin + (1 + nrows) * stride * 4);
		memcpy(hq, q, dim * 4);
		for (size_t d = dim; d < stride; d++) hq[d] = 0.f;
		for (size_t r = 0; r < nrows; r++)
		{
			memcpy(hr + r * stride, rows + r * dim, dim * 4);
			for (size_t d = dim; d < stride; d++) hr[r * stride + d] = 0.f;
		}
		int rc2 = hnsw_gpu_dist_batch_dev(func, hq, hr, nrows, dim, stride, ho, pin_stream);
		if (rc2) return rc2;
		HIPCHK(hipStreamSynchronize(pin_stream));
		memcpy(out, ho, nrows * 4);
		return HNSW_GPU_OK;
	}
	float *dq = nullptr, *dr = nullptr, *dout = nullptr;
	hipError_t e = hipSuccess;
	int rc = HNSW_GPU_OK;
	if ((e = hipMalloc(&dq, dim * 4)) != hipSuccess || (e = hipMalloc(&dr, nrows * stride * 4)) != hipSuccess ||
		(e = hipMalloc(&dout, nrows * 4)) != hipSuccess)
		rc = fail(HNSW_GPU_ERR_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
	if (!rc && stride != dim && (e = hipMemset(dr, 0, nrows * stride * 4)) != hipSuccess) rc = fail(HNSW_GPU_ERR_HIP, "memset failed");
	if (!rc && ((e = hipMemcpy(dq, q, dim * 4, hipMemcpyHostToDevice)) != hipSuccess ||
				(e = hipMemcpy2D(dr, stride * 4, rows, dim * 4, dim * 4, nrows, hipMemcpyHostToDevice)) != hipSuccess))
		rc = fail(HNSW_GPU_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
	if (!rc) rc = hnsw_gpu_dist_batch_dev(func, dq, dr, nrows, dim, stride, dout, nullptr);
	if (!rc && (e = hipMemcpy(out, dout, nrows * 4, hipMemcpyDeviceToHost)) != hipSuccess)
		rc = fail(HNSW_GPU_ERR_HIP, "download failed: %s", hipGetErrorString(e));
	if (dq) (void) hipFree(dq);
	if (dr) (void) hipFree(dr);
	if (dout) (void) hipFree(dout);
	return rc;
}

// ------------------------------------------------------------------------------------
// exhaustive k-NN with the same distance code (recall ground truth)
// ------------------------------------------------------------------------------------
// grid = (splits, nq); each wave scans a contiguous slice of the rows for one query and keeps a
// sorted top-k of (ord(dist)<<32 | idx) keys in LDS; partial lists are merged by topk_merge_kernel.
template <int FUNC>
__global__ __launch_bounds__(256) void bruteforce_kernel(const float *__restrict__ vec, uint32_t n, uint32_t dim,
														 uint32_t stride, uint32_t nchunks, uint32_t kiters,
														 uint32_t qpad_floats, const float *__restrict__ queries,
														 uint32_t k, uint64_t *__restrict__ part /* [nq][splits*4][k] */)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const uint32_t qi = blockIdx.y;
	float *qf = reinterpret_cast<float *>(smem);
	const float4 *q4 = reinterpret_cast<const float4 *>(smem);
	for (uint32_t e = threadIdx.x; e < qpad_floats; e += blockDim.x) qf[e] = (e < dim) ? queries[(size_t) qi * dim + e] : 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	uint64_t *top = reinterpret_cast<uint64_t *>(smem + (size_t) qpad_floats * 4) + (size_t) wib * (k + 1);
	float *sums = reinterpret_cast<float *>(smem + (size_t) qpad_floats * 4 + (size_t) 4 * (k + 1) * 8) + wib * 128;
	const uint32_t nw = gridDim.x * 4, w = blockIdx.x * 4 + wib;
	const uint32_t lo = (uint32_t) ((uint64_t) n * w / nw), hi = (uint32_t) ((uint64_t) n * (w + 1) / nw);
	float qnorm = 0.f;
	if (FUNC == F_COSINE) qnorm = query_norm(q4, nchunks, kiters, lane);
	uint32_t tsize = 0;
	uint64_t worst = ~0ull;
	for (uint32_t base = lo; base < hi; base += 64)
	{
		const uint32_t cnt = min(64u, hi - base);
		auto direct = [base](uint32_t r) { return base + r; };
		score_rows<FUNC, 4, 2>(vec, stride, q4, nchunks, kiters, direct, cnt, sums, lane);
		wave_sync();
		const float dl = finish_dist<FUNC>(sums[lane], sums[OUT2 + lane], qnorm);
		const uint64_t kl = ((uint64_t) ord_f32(dl) << 32) | (base + lane);
		// only rows that can enter the current top-k are visited one by one
		uint64_t todo = __ballot((uint32_t) lane < cnt && (tsize < k || kl < worst));
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
	uint64_t *dst = part + ((size_t) qi * nw + w) * k;
	for (uint32_t i = lane; i < k; i += 64) dst[i] = (i < tsize) ? top[i] : ~0ull;
}

// One wave per query: merge `nlists` ascending key lists of length k into the k smallest.
__global__ __launch_bounds__(64) void key_merge_kernel(const uint64_t *__restrict__ part, uint32_t nlists, uint32_t k,
													   uint32_t *__restrict__ out_idx, float *__restrict__ out_dist)
{
	const uint32_t qi = blockIdx.x;
	const int lane = threadIdx.x;
	const uint64_t *src = part + (size_t) qi * nlists * k;
	const uint32_t total = nlists * k;
	for (uint32_t x = lane; x < total; x += 64)
	{
		const uint32_t l = x / k;
		const uint64_t key = src[x];
		if (key == ~0ull) continue;
		uint32_t rank = x - l * k;
		for (uint32_t m = 0; m < nlists && rank < k; m++)
		{
			if (m == l) continue;
			const uint64_t *o = src + (size_t) m * k;
			uint32_t lo = 0, hi = k;                       // number of keys in list m below `key` (keys are unique)
			while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (o[mid] < key) lo = mid + 1; else hi = mid; }
			rank += lo;
		}
		if (rank < k)
		{
			out_idx[(size_t) qi * k + rank] = (uint32_t) key;
			if (out_dist) out_dist[(size_t) qi * k + rank] = unord_f32((uint32_t) (key >> 32));
		}
	}
}

__global__ void fill_u32_kernel(uint32_t *p, size_t n, uint32_t v)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

static int bruteforce_prefix(hnsw_gpu_index *ix, size_t nrows, const coord_t *d_queries, size_t nq, size_t k, idx_t *d_idx,
							 dist_t *d_dists, void *stream)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_queries || !d_idx) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (nq == 0) return HNSW_GPU_OK;
	if (k == 0 || k > 1024) return fail(HNSW_GPU_ERR_ARG, "k %zu out of range [1, 1024]", k);
	if (nq > 65535) return fail(HNSW_GPU_ERR_ARG, "at most 65535 queries per call");
	HIPCHK(hipSetDevice(ix->device));
	hipStream_t s = (hipStream_t) stream;
	const uint32_t nchunks = ix->stride / 4, kiters = (nchunks + 15) / 16;
	const uint32_t qpad = (uint32_t) round_up(kiters, 4) * 64;
	uint32_t splits = (uint32_t) std::max<size_t>(1, std::min<size_t>(64, (size_t) (4 * ix->num_cu) / nq));
	splits = (uint32_t) std::min<size_t>(splits, std::max<size_t>(1, nrows / 64));
	const uint32_t nlists = splits * 4;
	const size_t lds = (size_t) qpad * 4 + (size_t) 4 * (k + 1) * 8 + 4 * 128 * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "k/dim too large for brute force");
	int rc = ensure_scratch(ix, nq * nlists * k * 8);
	if (rc) return rc;
	uint64_t *part = (uint64_t *) ix->scratch;
	const size_t tot = nq * k;
	hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t) ((tot + 255) / 256)), dim3(256), 0, s, d_idx, tot, LINK_NONE);
	if (d_dists)
		hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t) ((tot + 255) / 256)), dim3(256), 0, s, (uint32_t *) d_dists, tot,
						   0x7F800000u);
	dim3 grid(splits, (uint32_t) nq);
#define BF_LAUNCH(F)                                                                                                   \
	hipLaunchKernelGGL(bruteforce_kernel<F>, grid, dim3(256), lds, s, ix->vec, (uint32_t) nrows, (uint32_t) ix->meta.dim, \
					   ix->stride, nchunks, kiters, qpad, d_queries, (uint32_t) k, part)
	switch ((int) ix->meta.dist_func)
	{
		case F_L2: BF_LAUNCH(F_L2); break;
		case F_COSINE: BF_LAUNCH(F_COSINE); break;
		default: BF_LAUNCH(F_MANHATTAN); break;
	}
#undef BF_LAUNCH
	hipLaunchKernelGGL(key_merge_kernel, dim3((uint32_t) nq), dim3(64), 0, s, part, nlists, (uint32_t) k, d_idx, d_dists);
	HIPCHK(hipGetLastError());
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_bruteforce_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k, idx_t *d_idx,
									   dist_t *d_dists, void *stream)
{
	if (!ix) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	return bruteforce_prefix(ix, ix->n, d_queries, nq, k, d_idx, d_dists, stream);
}

// ------------------------------------------------------------------------------------
// exhaustive k-NN with the dense part on the matrix cores (device_bf_mfma.h)
// ------------------------------------------------------------------------------------
static float g_last_bf_gemm_ms = 0.f;
static unsigned long long g_last_bf_clocks[2] = { 0, 0 };
static int g_last_bf_tile = 0;

// the filter launch for one tile shape (LDS per block: 69 KB for 128 x 128 tiles, 134 KB for 256 x 256; set per call: the attribute is per device)
template <int WM, int NJ>
static int bf_filter_launch(BfArgs &a, uint32_t nq, uint32_t n, hipStream_t s)
{
	using T = BfTile<WM, NJ>;
	a.nqt = (nq + T::TQ - 1) / T::TQ;
	a.nrt = (n + T::TR - 1) / T::TR;
	const uint32_t rgroups = (a.nrt + 7) / 8;
	HIPCHK(hipFuncSetAttribute((const void *) bf_mfma_filter_kernel<WM, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) T::LDS_BYTES));
	hipLaunchKernelGGL((bf_mfma_filter_kernel<WM, NJ>), dim3(rgroups * a.nqt * 8), dim3(T::THREADS), T::LDS_BYTES, s, a);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_bruteforce_mfma_dev(hnsw_gpu_index *ix, const coord_t *d_queries, size_t nq, size_t k,
											idx_t *d_idx, dist_t *d_dists, void *stream_)
{
	std::unique_lock<std::recursive_mutex> lock_;
	if (ix) lock_ = std::unique_lock<std::recursive_mutex>(ix->mu);
	if (!ix || !d_queries || !d_idx) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (nq == 0) return HNSW_GPU_OK;
	if (k == 0 || k > 1024) return fail(HNSW_GPU_ERR_ARG, "k %zu out of range [1, 1024]", k);
	if (nq > 65535) return fail(HNSW_GPU_ERR_ARG, "at most 65535 queries per call");
	const int func = (int) ix->meta.dist_func;
	if (func == F_MANHATTAN || ix->n < 4096)          // not a contraction / too small to matter
		return hnsw_gpu_bruteforce_dev(ix, d_queries, nq, k, d_idx, d_dists, stream_);
	HIPCHK(hipSetDevice(ix->device));
	{
		// The filter kernel is written for gfx950: 16-byte direct-to-LDS loads and 69 / 134 KB of LDS per block.  Anything else gets the
		// canonical scan — the same answer, bit for bit (the filter's survivors are re-scored by that code anyway).
		if (!ix->gfx950 || ix->max_lds < (size_t) 72 * 1024)
			return hnsw_gpu_bruteforce_dev(ix, d_queries, nq, k, d_idx, d_dists, stream_);
	}
	hipStream_t s = (hipStream_t) stream_;
	const uint32_t n = (uint32_t) ix->n, stride = ix->stride, dim = (uint32_t) ix->meta.dim;
	const uint32_t nchunks = stride / 4, kiters = (nchunks + 15) / 16;

	// |row|^2 cache
	if (ix->xnorm_cap < ix->n)
	{
		if (ix->xnorm) (void) hipFree(ix->xnorm);
		ix->xnorm = nullptr; ix->xnorm_cap = 0; ix->xnorm_n = 0;
		HIPCHK(hipMalloc(&ix->xnorm, ix->cap * sizeof(float)));
		ix->xnorm_cap = ix->cap;
	}
	if (ix->xnorm_n != ix->n)
	{
		hipLaunchKernelGGL(row_norm2_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix->vec, n, stride, ix->xnorm);
		ix->xnorm_n = ix->n;
	}

	const uint32_t cap = 16384;
	const size_t sample = std::min<size_t>(ix->n, std::max<size_t>(8192, (size_t) k * ix->n / 2048));
	// scratch carve
	const size_t o_q = 0;
	const uint32_t qstride = (uint32_t) round_up(stride, BF_TK);        // the filter's query copy: zero padded to whole K steps
	const size_t o_qn = o_q + round_up(nq * qstride * 4, 256);
	const size_t o_sidx = o_qn + round_up(nq * 4, 256);
	const size_t o_sdist = o_sidx + round_up(nq * k * 4, 256);
	const size_t o_bound = o_sdist + round_up(nq * k * 4, 256);
	const size_t o_cnt = o_bound + round_up(nq * 4, 256);
	const size_t o_cand = o_cnt + round_up(nq * 4 + 64, 256);
	const size_t o_clk = o_cand + round_up(nq * (size_t) cap * 4, 256);
	const size_t total = o_clk + 256;
	if (total > ix->bf_bytes)
	{
		if (ix->bf) (void) hipFree(ix->bf);
		ix->bf = nullptr; ix->bf_bytes = 0;
		HIPCHK(hipMalloc(&ix->bf, total));
		ix->bf_bytes = total;
	}
	char *B = (char *) ix->bf;
	float *qpad = (float *) (B + o_q), *qn = (float *) (B + o_qn), *sdist = (float *) (B + o_sdist), *bound = (float *) (B + o_bound);
	uint32_t *sidx = (uint32_t *) (B + o_sidx), *cnt = (uint32_t *) (B + o_cnt), *cand = (uint32_t *) (B + o_cand);
	uint32_t *overflow = cnt + nq;

	// 1. bound per query from a canonical scan of the sample rows
	int rc = bruteforce_prefix(ix, sample, d_queries, nq, k, sidx, sdist, s);
	if (rc) return rc;
	const size_t qtot = nq * (size_t) qstride;
	hipLaunchKernelGGL(pad_queries_kernel, dim3((uint32_t) ((qtot + 255) / 256)), dim3(256), 0, s, d_queries, (uint32_t) nq, dim, qstride, qpad);
	hipLaunchKernelGGL(row_norm2_kernel, dim3((uint32_t) ((nq + 3) / 4)), dim3(256), 0, s, qpad, (uint32_t) nq, qstride, qn);
	// tau_q = sdist[q*k + k-1]: gather with a strided view
	{
		// reuse make_bounds on a compacted tau array: write tau into `bound` first
		hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(1), 0, s, overflow, (size_t) 1, 0u);
		HIPCHK(hipMemcpy2DAsync(bound, 4, sdist + (k - 1), k * 4, 4, nq, hipMemcpyDeviceToDevice, s));
		hipLaunchKernelGGL(make_bounds_kernel, dim3((uint32_t) ((nq + 255) / 256)), dim3(256), 0, s, bound, qn, (uint32_t) nq, func, bound);
	}
	HIPCHK(hipMemsetAsync(cnt, 0, nq * 4, s));

	// 2. the dense contraction + filter
	BfArgs a;
	memset(&a, 0, sizeof(a));
	a.queries = qpad; a.qnorm = qn; a.qbound = bound; a.vec = ix->vec; a.xnorm = ix->xnorm;
	a.nq = (uint32_t) nq; a.n = n; a.stride = stride; a.qstride = qstride; a.ksteps = qstride / BF_TK; a.func = func;
	a.cand = cand; a.cand_cnt = cnt; a.cap = cap; a.clocks = (unsigned long long *) (B + o_clk);
	HIPCHK(hipMemsetAsync(a.clocks, 0, 16, s));                  // (written only by a block from the middle of the launch that does not exit early: a small table must not leave stale ticks behind)
	if (!ix->bf_e0) { HIPCHK(hipEventCreate(&ix->bf_e0)); HIPCHK(hipEventCreate(&ix->bf_e1)); }
	hipEvent_t e0 = ix->bf_e0, e1 = ix->bf_e1;
	HIPCHK(hipEventRecord(e0, s));
	// 256 x 256 tiles when they compute no more padding than 128 x 128 tiles would (an even number of 128-query tiles) and there are
	// tiles enough to fill the device several times over; the same dot products in the same k order either way: the same survivors
	{
		using Big = BfTile<4, 4>;
		const uint64_t nqt_s = (nq + BfTile<BF_WM, BF_NJ>::TQ - 1) / BfTile<BF_WM, BF_NJ>::TQ;
		const uint64_t big_blocks = ((nq + Big::TQ - 1) / Big::TQ) * ((n + Big::TR - 1) / Big::TR);
		// (test knob: 0 = never, < 0 = always, n = at least n blocks; the tests run every case through both tiles)
		const long long min_blocks = knob(K_BF_BIG_MIN_BLOCKS, 2048);
		const bool big = BF_BIG && BF_WM == 2 && BF_NJ == 2 && min_blocks != 0 &&
						 (min_blocks < 0 || (nqt_s % 2 == 0 && big_blocks >= (uint64_t) min_blocks));
		rc = big ? bf_filter_launch<4, 4>(a, (uint32_t) nq, n, s) : bf_filter_launch<BF_WM, BF_NJ>(a, (uint32_t) nq, n, s);
		if (rc) return rc;
		g_last_bf_tile = big ? Big::TQ : BfTile<BF_WM, BF_NJ>::TQ;
	}
	HIPCHK(hipEventRecord(e1, s));

	// 3. canonical re-score of the survivors
	const uint32_t qpadf = (uint32_t) round_up(kiters, 4) * 64;
	const size_t wave_bytes = round_up((size_t) qpadf * 4 + (k + 1) * 8 + 128 * 4, 16);
	const size_t lds = wave_bytes * 4;
	if (lds > 64 * 1024) return fail(HNSW_GPU_ERR_ARG, "k/dim too large for the rescoring step");
#define RS_LAUNCH(F)                                                                                                      \
	hipLaunchKernelGGL(bf_rescore_kernel<F>, dim3((uint32_t) ((nq + 3) / 4)), dim3(256), lds, s, ix->vec, dim, stride,      \
					   nchunks, kiters, qpadf, d_queries, (uint32_t) nq, cand, cnt, cap, (uint32_t) k, d_idx, d_dists, overflow)
	if (func == F_L2) RS_LAUNCH(F_L2); else RS_LAUNCH(F_COSINE);
#undef RS_LAUNCH
	HIPCHK(hipGetLastError());
	uint32_t ovf = 0;
	HIPCHK(hipMemcpyAsync(&ovf, overflow, 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(g_last_bf_clocks, a.clocks, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	(void) hipEventElapsedTime(&g_last_bf_gemm_ms, e0, e1);
	if (ovf)      // a candidate list overflowed (bound far too loose for some query): canonical scan instead
		return hnsw_gpu_bruteforce_dev(ix, d_queries, nq, k, d_idx, d_dists, stream_);
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_last_bruteforce_tile(void) { return g_last_bf_tile; }

/* device time of the MFMA filter kernel of the most recent hnsw_gpu_bruteforce_mfma_dev call */
extern "C" float hnsw_gpu_last_bruteforce_gemm_ms(void) { return g_last_bf_gemm_ms; }

/* shader-clock MHz during that kernel: ticks of the shader clock over ticks of the constant 100 MHz clock, both taken by block 0
 * around its K loop — what the matrix roof has to be priced at when the device does not hold its nominal clock under this load */
extern "C" double hnsw_gpu_last_bruteforce_clock_mhz(void)
{
	int khz = 0, dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
		khz = 100000;
	return g_last_bf_clocks[1] ? khz * 1e-3 * (double) g_last_bf_clocks[0] / (double) g_last_bf_clocks[1] : 0.0;
}

