// embedding_shim.cpp — libembedding_gpu.so: the four symbols embedding.c links against
// (embedding.h:46-47,55-56), implemented on the MI355X library (libhnsw_gpu.so).
//
//   hnsw_search          hnswalg.cpp:256-277  -> one-query launch of the fused search kernel
//   hnsw_dist_func       distfunc.c:171-174   -> one pair, computed on the calling core in the device's
//                                                canonical summation order (host_dist.h): bit-identical to
//                                                hnsw_gpu_dist_batch at ~0.1 us instead of a 16-28 us launch
//   hnsw_init_dist_func  distfunc.c:159-169   -> device selection / runtime warm-up
//   hnsw_bind_point      hnswalg.cpp:279-291  -> serial device insert + write-back of the changed
//                                                link lists through hnsw_begin_write
//
// It imports the host's storage callbacks (embedding.h:44,48-53) exactly like hnswalg.cpp does.
// There is no CPU implementation of the search or the insert: without a gfx950 device hnsw_search and
// hnsw_bind_point fail (false) and say why on stderr.  hnsw_dist_func is host code by design (its only
// caller left is the SQL operator, one pair per call, embedding.c:1037), not a fallback.
//
// Where does the index come from?  HnswMetadata carries no relation identity and Postgres
// rebuilds it for every scan (embedding.c:254).  A mirror is built by walking the host's elements
// through hnsw_begin_read/hnsw_end_read (one pin at a time: the host allows at most 4,
// embedding.c:40,714-715).  Without further help from the host it is kept across calls and every
// answer is validated against the host's pages along the walk that produced it (shim_cache.h;
// PG_EMBEDDING_GPU_CACHE=0: a fresh mirror per call, O(N), the round-1 behaviour).  A host that
// knows its index is unchanged attaches a mirror once (hnsw_gpu_shim_attach) and every later search
// through the drop-in symbol is a single kernel launch — see INTEGRATION.md.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "hnsw_gpu.h"
#include "hnsw_gpu_shim.h"
#include "host_walk.h"
#include "host_dist.h"
#include "shim_cache.h"

namespace {

std::mutex g_mu;
int g_device = -1;                 // chosen by hnsw_init_dist_func

// attached mirrors by the address of the host's HnswMetadata (one per open scan / insert in the host); any number
std::unordered_map<HnswMetadata *, hnsw_gpu_index *> *g_attached = nullptr;     // (leaked on purpose: no static destructor order issues at exit)

int pick_device()
{
	if (g_device >= 0) return g_device;
	const char *env = getenv("PG_EMBEDDING_GPU_DEVICE");
	int dev = env ? atoi(env) : 0;
	int n = hnsw_gpu_device_count();
	if (n <= 0) return -1;
	if (dev < 0 || dev >= n) dev = 0;
	g_device = dev;
	return dev;
}

hnsw_gpu_index *find_attached(HnswMetadata *meta)
{
	if (!g_attached) return nullptr;
	auto it = g_attached->find(meta);
	return it == g_attached->end() ? nullptr : it->second;
}

// Staging buffer for the callback walk.  Static and reused so that a host callback which
// leaves by longjmp (elog(ERROR), embedding.c:715) cannot leak it.
thread_local char  *t_stage = nullptr;
thread_local size_t t_stage_bytes = 0;

bool stage_reserve(size_t bytes)
{
	if (bytes <= t_stage_bytes) return true;
	size_t nb = t_stage_bytes ? t_stage_bytes : (1u << 20);
	while (nb < bytes) nb *= 2;
	char *p = (char *) realloc(t_stage, nb);
	if (!p) return false;
	memset(p + t_stage_bytes, 0, nb - t_stage_bytes);
	t_stage = p;
	t_stage_bytes = nb;
	return true;
}

}  // namespace

// Copy the host index into one contiguous image by the same accessor the reference search uses
// (hnsw_begin_read, embedding.c:704-757), following the links from the entry point — see
// host_walk.h for why it must not probe element numbers past the end (the real host raises ERROR
// there) and why leaving out unreachable elements changes no answer.  Element numbers that were
// not reached, among them the holes at the tail of a page (idx = blk*elems_per_page + off-1 with
// un-aligned elems_per_page, embedding.c:229,693 / SURVEY.md §0.8), become zero-linked,
// vacuum-flagged placeholders that nothing links to.
static int hnsw_gpu_shim_snapshot_impl(HnswMetadata *meta, hnsw_gpu_index **out)
{
	if (!meta || !out) return HNSW_GPU_ERR_ARG;
	int dev = pick_device();
	if (dev < 0)
	{
		fprintf(stderr, "pg_embedding_amd: no HIP device visible; the GPU hot path has no CPU fallback\n");
		return HNSW_GPU_ERR_NODEVICE;
	}
	const long n = hostwalk::copy_reachable(meta, [](size_t bytes) -> char * { return stage_reserve(bytes) ? t_stage : nullptr; });
	if (n < 0) return HNSW_GPU_ERR_NOMEM;
	return hnsw_gpu_index_create_from_flat(meta, t_stage, (size_t) n, dev, out);
}

extern "C" int hnsw_gpu_shim_attach(HnswMetadata *meta, hnsw_gpu_index *ix)
{
	if (!meta || !ix) return HNSW_GPU_ERR_ARG;
	std::lock_guard<std::mutex> lk(g_mu);
	try
	{
		if (!g_attached) g_attached = new std::unordered_map<HnswMetadata *, hnsw_gpu_index *>();
		(*g_attached)[meta] = ix;
	}
	catch (...) { return HNSW_GPU_ERR_NOMEM; }
	return HNSW_GPU_OK;
}

extern "C" int hnsw_gpu_shim_detach(HnswMetadata *meta)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_attached && g_attached->erase(meta)) return HNSW_GPU_OK;
	return HNSW_GPU_ERR_ARG;
}

// ---------------------------------------------------------------------------------------
// the drop-in symbols
// ---------------------------------------------------------------------------------------

extern "C" void hnsw_init_dist_func(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	(void) pick_device();
}

extern "C" dist_t hnsw_dist_func(dist_func_t dist, coord_t const *ax, coord_t const *bx, size_t dim)
{
	if (!ax || !bx) return NAN;
	return hostdist::dist((int) dist, ax, bx, dim);
}

static bool hnsw_search_impl(HnswMetadata *meta, const coord_t *point, size_t *n_results, label_t **results)
{
	if (!meta || !point || !n_results || !results) return false;
	const size_t ef = meta->efSearch;
	hnsw_gpu_index *ix = nullptr;
	bool own = false;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		ix = find_attached(meta);
	}
	if (ef == 0)
	{
		// searchKnn trims to k = 0 results (hnswalg.cpp:238-240): nothing to mirror or launch for that
		label_t *none = (label_t *) malloc(1);
		if (!none) return false;
		*n_results = 0;
		*results = none;
		return true;
	}
	if (!ix && shimcache::enabled())
	{
		// no mirror attached: the validated cache (shim_cache.h) — a mirror kept across calls, every answer checked
		// against the host's pages along the walk that produced it
		const int dev = pick_device();
		if (dev < 0)
		{
			fprintf(stderr, "pg_embedding_amd: no HIP device visible; the GPU hot path has no CPU fallback\n");
			return false;
		}
		// The walk runs host callbacks (hnsw_begin_read) that may leave by longjmp (elog(ERROR)): nothing malloc'd is held
		// across it — the results land in a per-thread buffer (reclaimed by the next call, like the walk's other buffers) and
		// the caller's array is allocated after the walk, as the reference does (hnswalg.cpp:262).
		static thread_local std::vector<label_t> tl_res;
		try { if (tl_res.size() < ef) tl_res.resize(ef); } catch (...) { return false; }
		uint32_t cnt = 0;
		if (shimcache::search(meta, point, ef, tl_res.data(), &cnt, dev))
		{
			label_t *cbuf = (label_t *) malloc((cnt ? cnt : 1) * sizeof(label_t));   // caller frees (embedding.c:327)
			if (!cbuf) return false;
			memcpy(cbuf, tl_res.data(), (size_t) cnt * sizeof(label_t));
			*n_results = cnt;
			*results = cbuf;
			return true;
		}
		fprintf(stderr, "pg_embedding_amd: hnsw_search failed: %s\n", hnsw_gpu_last_error());
		return false;
	}
	if (!ix)
	{
		if (hnsw_gpu_shim_snapshot(meta, &ix) != HNSW_GPU_OK)
		{
			fprintf(stderr, "pg_embedding_amd: hnsw_search: cannot mirror the index: %s\n", hnsw_gpu_last_error());
			return false;
		}
		own = true;
	}
	bool ok = false;
	label_t *buf = (label_t *) malloc(ef * sizeof(label_t));   // caller frees (embedding.c:327)
	uint32_t cnt = 0;
	if (buf && hnsw_gpu_search_batch(ix, point, 1, ef, buf, nullptr, &cnt) == HNSW_GPU_OK)
	{
		*n_results = cnt;
		ok = true;
	}
	else
		fprintf(stderr, "pg_embedding_amd: hnsw_search failed: %s\n", hnsw_gpu_last_error());
	if (own) hnsw_gpu_index_destroy(ix);
	if (!ok) { free(buf); return false; }
	*results = buf;
	return true;
}

// hnsw_bind_point (hnswalg.cpp:279-291).  The host has already stored element `idx`
// zero-linked (embedding.c:619-621,670).  The device runs the reference's insert in its serial
// form (hnsw_gpu_index_link, max_batch = 1: searchBaseLayer(efConstruction) +
// mutuallyConnectNewElement + getNeighborsByHeuristic, bit-identical link lists), then the
// changed lists — the new element's and those of the elements it linked to — are written back
// to the host through hnsw_begin_write/hnsw_end_write (one write pin at a time,
// embedding.c:780-781).  With an attached mirror that already holds elements [0, idx) only the
// new element is uploaded; otherwise the index is mirrored first.
// Where an insert's time goes (cumulative ns per thread; hnsw_gpu_shim_insert_times): the validated cache's preparation (pick + the
// traced validation walk with the host-side comparison), the device insert (hnsw_gpu_index_insert_one or link + gather), the
// write-back through hnsw_begin_write, and everything else (label read, reserve, shadow upkeep).
static thread_local uint64_t t_ins[5] = {0, 0, 0, 0, 0};     // prepare, device, write-back, other, calls
static inline uint64_t now_ns()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t) ts.tv_sec * 1000000000ull + (uint64_t) ts.tv_nsec;
}

static bool hnsw_bind_point_impl(HnswMetadata *meta, const coord_t *point, idx_t idx)
{
	const uint64_t t_enter = now_ns();
	uint64_t t_prep = 0, t_dev = 0, t_wb = 0;
	if (!meta || !point) return false;
	hnsw_gpu_index *ix = nullptr;
	bool own = false;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		ix = find_attached(meta);
	}
	// bindPoint links nothing for the first element (hnswalg.cpp:228); an attached mirror still has to
	// receive the row itself.
	if (idx == 0 && !ix) return true;
	const size_t maxM = meta->maxM;
	if (maxM > 4096)                                     // (the mirror itself refuses such an index too: check_meta)
	{
		fprintf(stderr, "pg_embedding_amd: hnsw_bind_point: maxM = %zu is not supported (at most 4096, i.e. m <= 2048)\n", maxM);
		return false;
	}
	static thread_local idx_t mine[4097];
	static thread_local std::vector<idx_t> others;            // the neighbours' lists, all fetched in one launch
	bool ok = false;
	bool fused = false;                                  // append + link + gather went as one call (hnsw_gpu_index_insert_one)
	shimcache::Entry *ce = nullptr;                      // cached mirror (unmodified glue): its shadow follows the insert
	do
	{
		if (!ix && shimcache::enabled())
		{
			const int dev = pick_device();
			if (dev < 0) { fprintf(stderr, "pg_embedding_amd: no HIP device visible; the GPU hot path has no CPU fallback\n"); break; }
			const uint64_t tp = now_ns();
			ce = shimcache::prepare_insert(meta, point, idx, dev);   // every element the insert will read == the host's
			t_prep = now_ns() - tp;
			if (!ce) break;
			ix = ce->ix;
			ce->suspect = true;                          // until the write-back below has completed
		}
		if (!ix)                                         // no attached mirror: mirror what the search can reach;
		{                                                // the new element is not linked yet, so it is added below
			if (hnsw_gpu_shim_snapshot(meta, &ix) != HNSW_GPU_OK) break;
			own = true;
		}
		{
			size_t have = hnsw_gpu_index_count(ix);
			if (have < (size_t) idx)                     // element numbers skipped a page-tail hole (embedding.c:229,693)
			{                                            // or the tail was not reachable: dead placeholders
				const size_t gap = (size_t) idx - have;
				if ((size_t) idx + 1 > hnsw_gpu_index_capacity(ix) &&          // (grow by half, and only when the row does not fit: a reserve
					hnsw_gpu_index_reserve(ix, (size_t) idx + 1 + (size_t) idx / 2) != HNSW_GPU_OK) break;   //  is a copy of the whole mirror — round 2 asked for it at every insert)
				coord_t *zeros = (coord_t *) calloc(gap * meta->dim, sizeof(coord_t));
				label_t *dead = (label_t *) malloc(gap * sizeof(label_t));
				bool fine = zeros && dead;
				for (size_t g = 0; fine && g < gap; g++) dead[g] = (label_t) 1 << HNSW_LABEL_DELETED_BIT;
				fine = fine && hnsw_gpu_index_append(ix, zeros, dead, gap) == HNSW_GPU_OK;
				free(zeros); free(dead);
				if (!fine) break;
				have = (size_t) idx;
			}
			if (have == (size_t) idx)                    // mirror is exactly one element behind: add it
			{
				label_t label = 0;
				if (!hnsw_begin_read(meta, idx, nullptr, nullptr, &label)) break;
				hnsw_end_read(meta);
				if ((size_t) idx + 1 > hnsw_gpu_index_capacity(ix) &&          // (grow by half, and only when the row does not fit: a reserve
					hnsw_gpu_index_reserve(ix, (size_t) idx + 1 + (size_t) idx / 2) != HNSW_GPU_OK) break;   //  is a copy of the whole mirror — round 2 asked for it at every insert)
				// the row, its links and the changed lists in one call: nothing waits on the host between the steps
				others.resize(maxM * (maxM + 1));
				const uint64_t td = now_ns();
				const shimcache::InsertWalk &iw = shimcache::insert_walk();
				// (under the validated cache the walk for this point has just been done, to CHECK the mirror: its result is the
				// insert's candidate list — the same query on the same graph — so the device does not walk a second time)
				const int irc = (ce && iw.valid && !getenv("PG_EMBEDDING_GPU_INSERT_REWALK"))
					? hnsw_gpu_index_insert_candidates(ix, point, label, idx, iw.idx.data(), iw.dist.data(), iw.cnt, mine, others.data())
					: hnsw_gpu_index_insert_one(ix, point, label, idx, mine, others.data());
				if (irc != HNSW_GPU_OK) break;
				t_dev = now_ns() - td;
				fused = true;
				if (ce && !shimcache::shadow_append(meta, ce, (size_t) idx + 1, idx, point, label)) break;
			}
			else if (have != (size_t) idx + 1)
			{
				fprintf(stderr, "pg_embedding_amd: hnsw_bind_point(%u): attached mirror holds %zu elements\n",
						(unsigned) idx, have);
				break;
			}
		}
		if (idx == 0) { ok = true; break; }
		if (!fused)                                          // the mirror already held the row (an attached mirror the host appended to)
		{
			if (hnsw_gpu_index_link(ix, idx, 1, 1, 0, nullptr) != HNSW_GPU_OK) break;
			others.resize(maxM * (maxM + 1));
			if (hnsw_gpu_index_get_link_lists(ix, idx, mine, others.data()) != HNSW_GPU_OK) break;
		}
		const uint64_t tw = now_ns();
		for (uint32_t j = 0; j < mine[0]; j++)               // neighbours first, like hnswalg.cpp:183-222 ...
		{
			const idx_t *other = others.data() + (size_t) j * (maxM + 1);
			idx_t *dst = nullptr;
			hnsw_begin_write(meta, mine[1 + j], &dst, nullptr, nullptr);
			memcpy(dst, other, (maxM + 1) * sizeof(idx_t));
			hnsw_end_write(meta);
			if (ce) shimcache::shadow_set_links(meta, ce, mine[1 + j], other);
		}
		{                                                   // ... then the element itself (:169-181)
			idx_t *dst = nullptr;
			hnsw_begin_write(meta, idx, &dst, nullptr, nullptr);
			memcpy(dst, mine, (maxM + 1) * sizeof(idx_t));
			hnsw_end_write(meta);
			if (ce) shimcache::shadow_set_links(meta, ce, idx, mine);
		}
		t_wb = now_ns() - tw;
		ok = true;
	} while (0);
	if (ce)
	{
		if (ok && !ce->ephemeral) ce->suspect = false;
		else shimcache::drop(ce);                        // mirror and host may have parted (or the index is too large to keep): never reuse it
	}
	if (!ok)
		fprintf(stderr, "pg_embedding_amd: hnsw_bind_point(%u) failed: %s\n", (unsigned) idx, hnsw_gpu_last_error());
	if (own && ix) hnsw_gpu_index_destroy(ix);
	{
		const uint64_t total = now_ns() - t_enter;
		t_ins[0] += t_prep; t_ins[1] += t_dev; t_ins[2] += t_wb; t_ins[3] += total - t_prep - t_dev - t_wb; t_ins[4]++;
	}
	return ok;
}

// ---------------------------------------------------------------------------------------
// The C boundary.  Nothing may unwind through it (the callers are C: embedding.c): an allocation
// failure inside the library (std::bad_alloc) becomes an ordinary failure, as the reference turns
// every exception into `false` at the same place (hnswalg.cpp:258-276, 281-290).
// ---------------------------------------------------------------------------------------
extern "C" void hnsw_gpu_shim_cache_stats(uint64_t out[8])
{
	const shimcache::Stats &s = shimcache::stats();
	out[0] = s.snapshots; out[1] = s.searches; out[2] = s.search_rounds; out[3] = s.inserts; out[4] = s.insert_rounds;
	out[5] = s.patched; out[6] = s.fallbacks; out[7] = s.elements_read;
}
extern "C" void hnsw_gpu_shim_insert_times(uint64_t out[5])
{
	for (int i = 0; i < 5; i++) out[i] = t_ins[i];
}
extern "C" void hnsw_gpu_shim_cache_clear(void)
{
	try { while (!shimcache::table().empty()) shimcache::drop(shimcache::table().back()); } catch (...) {}
	shimcache::stats() = shimcache::Stats{};
}
extern "C" int hnsw_gpu_shim_snapshot(HnswMetadata *meta, hnsw_gpu_index **out) { try { return hnsw_gpu_shim_snapshot_impl(meta, out); } catch (...) { return HNSW_GPU_ERR_NOMEM; } }
extern "C" bool hnsw_search(HnswMetadata *meta, const coord_t *point, size_t *n_results, label_t **results) { try { return hnsw_search_impl(meta, point, n_results, results); } catch (...) { return false; } }
extern "C" bool hnsw_bind_point(HnswMetadata *meta, const coord_t *point, idx_t idx) { try { return hnsw_bind_point_impl(meta, point, idx); } catch (...) { return false; } }
