// shim_cache.h — libembedding_gpu.so's mirror cache for the UNMODIFIED glue (embedding.c as it is).
//
// The four-symbol boundary (embedding.h:46-47,55-56) tells the library neither which index a call is about
// nor whether it changed since the last call: HnswMetadata is rebuilt per scan (embedding.c:254), an INSERT
// rewrites link lists anywhere in the graph (hnswalg.cpp:183-222), a VACUUM flips flag bits inside labels
// (embedding.c:920-926).  No O(1) check through hnsw_begin_read can see all of that, so a
// mirror kept across calls is never trusted.  What IS checkable is the walk:
//
//   searchBaseLayer is a deterministic function of the elements it touches — the entry point, the elements it
//   expands (pops, hnswalg.cpp:73) and every link target of those.  The kernel reports its pop sequence
//   (hnsw_gpu_search_trace); the touched elements are read from the host through hnsw_begin_read — each one only
//   after an element that links to it was found byte-identical to the mirror's copy, so a read can never leave the
//   relation — and compared with the copy the mirror was built from.  All identical: the reference's own walk over
//   the host reads the same bytes in the same order and returns the same answer.  Any difference: the host's
//   images are patched into the mirror and the query runs again.
//
// Per query this reads through the host's buffer manager what the reference's search reads (E_q elements), plus
// the device walk: about twice the reference's latency, instead of a full O(N) re-mirror per call.  The fast forms
// remain the explicit attach (hnsw_gpu_shim_attach) and the server with the identity/generation patch.
//
// Memory: the cache keeps the flat image the mirror was built from (N x element size, host) beside the mirror (HBM), per
// THREAD (= per Postgres backend), for up to 4 indexes.  That is the price of the unmodified glue and it does not scale
// with connections: a host with many backends should run the server mode (one mirror per index for all of them,
// INTEGRATION.md §2).  Bounds: PG_EMBEDDING_GPU_CACHE_MAX_MB (default 4096) per index — larger ones are mirrored for the
// call that needs them and dropped again; entries idle for PG_EMBEDDING_GPU_CACHE_IDLE_S (default 300) are dropped at the
// next call into the library; an allocation failure while (re-)mirroring drops every other cached entry of the thread and
// tries once more before the call fails; an entry left half-patched by a longjmp is dropped at the next call.
// PG_EMBEDDING_GPU_CACHE=0 switches the cache off (every call re-mirrors, the round-1 behaviour).
// Not thread-shared: the table is thread_local (a Postgres backend is one thread; a threaded C host gets one cache
// per thread), so no lock is ever held across a host callback — a callback may leave by longjmp (elog(ERROR)).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "hnsw_gpu.h"
#include "hnsw_gpu_shim.h"
#include "host_walk.h"

namespace shimcache {

struct Stats { uint64_t snapshots, searches, search_rounds, inserts, insert_rounds, patched, fallbacks, elements_read; };
inline Stats &stats() { static thread_local Stats s = {}; return s; }

struct Entry
{
	HnswMetadata key;                  // the fields that define layout, build parameters and the metric
	hnsw_gpu_index *ix = nullptr;
	std::vector<char> shadow;          // the host images the mirror holds, elements [0, n)
	size_t n = 0;
	std::vector<uint32_t> stamp;       // 2*epoch = found identical in this validation, 2*epoch+1 = found different
	uint32_t epoch = 0;
	uint64_t last_use = 0;
	time_t last_time = 0;              // wall seconds of the last use (idle expiry)
	bool suspect = false;              // an insert was interrupted half-way (a callback left by longjmp)
	bool ephemeral = false;            // larger than the cache may keep: used for this call only, then dropped
};

inline std::vector<Entry *> &table() { static thread_local std::vector<Entry *> t; return t; }
inline uint64_t &clock_() { static thread_local uint64_t c = 0; return c; }

inline bool enabled()
{
	const char *e = getenv("PG_EMBEDDING_GPU_CACHE");
	return !(e && atoi(e) == 0);
}

// Host bytes one cached index may keep beside its mirror (the flat image it is validated against).  An index above
// the limit is mirrored for the call that needs it and dropped again — the round-1 behaviour, for that index only.
inline size_t max_shadow_bytes()
{
	const char *e = getenv("PG_EMBEDDING_GPU_CACHE_MAX_MB");
	const long long mb = e ? atoll(e) : 4096;
	return mb <= 0 ? 0 : (size_t) mb << 20;
}

inline bool same_key(const HnswMetadata &a, const HnswMetadata &b)
{
	return a.dim == b.dim && a.data_size == b.data_size && a.offset_data == b.offset_data && a.offset_label == b.offset_label &&
		   a.size_data_per_element == b.size_data_per_element && a.M == b.M && a.maxM == b.maxM &&
		   a.efConstruction == b.efConstruction && a.enterpoint_node == b.enterpoint_node && a.dist_func == b.dist_func;
}

inline void drop(Entry *e)
{
	auto &t = table();
	t.erase(std::remove(t.begin(), t.end(), e), t.end());
	if (e->ix) hnsw_gpu_index_destroy(e->ix);
	delete e;
}

inline time_t now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec;
}

inline long idle_limit_s()
{
	const char *e = getenv("PG_EMBEDDING_GPU_CACHE_IDLE_S");
	return e ? atol(e) : 300;
}

// Housekeeping at the start of every call: entries an interrupted insert left half-patched (they would never be picked
// again, but kept their HBM and host memory until LRU eviction) and entries nobody has used for a while.
inline void sweep()
{
	auto &t = table();
	const time_t now = now_s();
	const long lim = idle_limit_s();
	for (size_t i = 0; i < t.size();)
	{
		Entry *e = t[i];
		if (e->suspect || (lim > 0 && e->last_time && now - e->last_time > lim)) drop(e);      // (erases t[i])
		else i++;
	}
}

// Per-thread buffers that outlive a callback's longjmp (reclaimed by the next call).
struct Patches
{
	std::vector<uint32_t> ids;
	std::vector<char> images;
	void clear() { ids.clear(); images.clear(); }
	void add(uint32_t id, const void *img, size_t esz)
	{
		ids.push_back(id);
		const size_t o = images.size();
		images.resize(o + esz);
		memcpy(images.data() + o, img, esz);
	}
};
inline Patches &patches() { static thread_local Patches p; return p; }
inline std::vector<char> &walkbuf() { static thread_local std::vector<char> b; return b; }

// Full walk of the host index (host_walk.h) into `e` (a new entry when e is null); the mirror is rebuilt.
inline Entry *resnapshot(HnswMetadata *meta, Entry *e, int device)
{
	std::vector<char> &wb = walkbuf();
	const long n = hostwalk::copy_reachable(meta, [&wb](size_t bytes) -> char * {
		try { if (wb.size() < bytes) wb.resize(bytes); } catch (...) { return nullptr; }
		return wb.data();
	});
	if (n < 0) return nullptr;
	hnsw_gpu_index *ix = nullptr;
	int rc = hnsw_gpu_index_create_from_flat(meta, wb.data(), (size_t) n, device, &ix);
	if (rc == HNSW_GPU_ERR_NOMEM || rc == HNSW_GPU_ERR_HIP)
	{
		// out of HBM (or of pinned staging)?  this thread's own cache goes first: the mirror this call replaces, then every
		// other entry — a query that can be answered after giving up the cache must not fail
		if (e && e->ix) { hnsw_gpu_index_destroy(e->ix); e->ix = nullptr; }
		auto &all = table();
		for (size_t i = 0; i < all.size();)
		{
			if (all[i] != e) drop(all[i]);
			else i++;
		}
		rc = hnsw_gpu_index_create_from_flat(meta, wb.data(), (size_t) n, device, &ix);
	}
	if (rc != HNSW_GPU_OK)
	{
		if (e && !e->ix) drop(e);                                  // its mirror went in the attempt above
		std::vector<char>().swap(wb);
		return nullptr;
	}
	auto &t = table();
	if (!e)
	{
		if (t.size() >= 4)                                     // least recently used goes
		{
			Entry *old = t[0];
			for (Entry *x : t) if (x->last_use < old->last_use) old = x;
			drop(old);
		}
		e = new Entry();
		t.push_back(e);
	}
	else if (e->ix) hnsw_gpu_index_destroy(e->ix);
	e->key = *meta;
	e->ix = ix;
	e->n = (size_t) n;
	e->shadow.swap(wb);                                        // the walk's image becomes the shadow: no second copy of the index
	e->shadow.resize((size_t) n * meta->size_data_per_element);
	if (e->shadow.capacity() > e->shadow.size() + e->shadow.size() / 8)
	{
		try { e->shadow.shrink_to_fit(); } catch (...) {}           // (the walk's buffer grows in steps: give the slack back)
	}
	std::vector<char>().swap(wb);
	e->stamp.assign((size_t) n, 0);
	e->epoch = 0;
	e->suspect = false;
	e->ephemeral = e->shadow.size() > max_shadow_bytes();
	e->last_use = ++clock_();
	e->last_time = now_s();
	stats().snapshots++;
	return e;
}

// The entry whose copy of the entry point equals the host's, or null.  *empty = the host index has no entry point
// (hnsw_begin_read(entry) == false, hnswalg.cpp:56-57).
inline Entry *pick(HnswMetadata *meta, bool *empty)
{
	*empty = false;
	sweep();
	const size_t esz = meta->size_data_per_element;
	idx_t *links = nullptr;
	if (!hnsw_begin_read(meta, meta->enterpoint_node, &links, nullptr, nullptr)) { *empty = true; return nullptr; }
	// identity = the entry point's vector and heap TID (its link list and vacuum flag may change under the same index:
	// that is what the validation is for; a false match only costs a validation that fails)
	const char *himg = reinterpret_cast<const char *>(links);
	uint64_t hlab;
	memcpy(&hlab, himg + meta->offset_label, 8);
	Entry *hit = nullptr;
	for (Entry *e : table())
	{
		if (e->suspect || e->ephemeral || !same_key(e->key, *meta) || e->n <= meta->enterpoint_node) continue;
		const char *simg = e->shadow.data() + (size_t) meta->enterpoint_node * esz;
		uint64_t slab;
		memcpy(&slab, simg + meta->offset_label, 8);
		if (((hlab ^ slab) & 0xFFFFFFFFFFFFull) == 0 && memcmp(simg + meta->offset_data, himg + meta->offset_data, meta->dim * sizeof(coord_t)) == 0 &&
			(!hit || e->last_use > hit->last_use))
			hit = e;
	}
	hnsw_end_read(meta);
	if (hit) { hit->last_use = ++clock_(); hit->last_time = now_s(); }
	return hit;
}

// Compare with the host every element a walk touches: the entry point and all link targets of the expanded elements that
// were themselves found identical.  Incremental, so that the comparison runs WHILE the device walks (the expanded elements
// arrive through hnsw_gpu_search_trace_poll as the kernel reports them).  Differences are collected in patches() (host
// images).  `missing`: the host no longer has an element the mirror's graph names (another index behind the same
// parameters: the caller re-mirrors).
struct Validator
{
	HnswMetadata *meta;
	Entry *e;
	size_t esz, maxM;
	uint32_t clean, dirty;
	bool missing = false;

	Validator(HnswMetadata *m, Entry *en) : meta(m), e(en), esz(m->size_data_per_element), maxM(m->maxM)
	{
		patches().clear();
		if (++e->epoch >= 0x7FFFFFF0u) { std::fill(e->stamp.begin(), e->stamp.end(), 0u); e->epoch = 1; }
		clean = 2 * e->epoch; dirty = clean + 1;
		visit(meta->enterpoint_node);
	}
	void visit(uint32_t id)
	{
		if ((size_t) id < e->stamp.size() && (e->stamp[id] == clean || e->stamp[id] == dirty)) return;
		if ((size_t) id >= e->stamp.size()) e->stamp.resize((size_t) id + 1 + e->stamp.size() / 4, 0);
		idx_t *links = nullptr;
		if (!hnsw_begin_read(meta, (idx_t) id, &links, nullptr, nullptr)) { missing = true; e->stamp[id] = dirty; return; }
		stats().elements_read++;
		const bool same = (size_t) id < e->n && memcmp(e->shadow.data() + (size_t) id * esz, links, esz) == 0;
		if (!same) patches().add(id, links, esz);              // the element image is contiguous from its link count on
		hnsw_end_read(meta);
		e->stamp[id] = same ? clean : dirty;
	}
	void expand(uint32_t x)
	{
		if (missing || (size_t) x >= e->stamp.size() || e->stamp[x] != clean) return;   // never follow links that are not the host's
		const uint32_t *l = reinterpret_cast<const uint32_t *>(e->shadow.data() + (size_t) x * esz);
		const uint32_t cnt = l[0] <= maxM ? l[0] : (uint32_t) maxM;
		// the copies to compare against lie scattered over an image of the whole index: start their cache misses together
		for (uint32_t j = 1; j <= cnt; j++)
			if ((size_t) l[j] < e->n)
			{
				const char *img = e->shadow.data() + (size_t) l[j] * esz;
				__builtin_prefetch(img);
				__builtin_prefetch(img + 64);
				__builtin_prefetch(img + (esz > 256 ? 256 : 0));
			}
		for (uint32_t j = 1; j <= cnt && !missing; j++) visit(l[j]);
	}
	long result() const { return missing ? -1 : (long) patches().ids.size(); }
};

// One traced walk on the mirror, validated while it runs.  `extra_results`: the results are expanded too (inserts).
// Returns the number of differing elements (0 = the answer in labels/count is the host's), -1 = re-mirror, -2 = failure.
inline long traced_walk(HnswMetadata *meta, Entry *e, const coord_t *point, size_t ef, int base, label_t *labels, uint32_t *count,
						std::vector<uint32_t> &pops, size_t cap, bool expand_results, dist_t *dists = nullptr)
{
	if (hnsw_gpu_search_trace_begin(e->ix, point, ef, base, cap) != HNSW_GPU_OK) return -2;
	Validator v(meta, e);                                       // (host callbacks from here on: no library lock is held)
	size_t have = 0;
	for (;;)
	{
		size_t got = 0;
		int finished = 0;
		if (hnsw_gpu_search_trace_poll(e->ix, pops.data() + have, cap - have, &got, &finished) != HNSW_GPU_OK) return -2;
		for (size_t i = 0; i < got; i++) v.expand(pops[have + i]);
		have += got;
		if (finished) break;
		if (got == 0) __builtin_ia32_pause();
	}
	uint32_t npops = 0;
	if (hnsw_gpu_search_trace_end(e->ix, labels, dists, count, &npops, nullptr) != HNSW_GPU_OK) return -2;
	if (npops > cap || have != npops) return -1;                // a walk too long to validate
	if (expand_results)
	{
		if ((size_t) npops + *count > pops.size()) return -1;
		for (uint32_t i = 0; i < *count; i++) { pops[have + i] = (uint32_t) labels[i]; v.expand(pops[have + i]); }
		have += *count;
	}
	const long diff = v.result();
	if (diff == 0)
	{
		const uint32_t ok = 2 * e->epoch;
		for (size_t i = 0; i < have; i++)
			if ((size_t) pops[i] >= e->stamp.size() || e->stamp[pops[i]] != ok) return -1;   // cannot happen after a clean validation
	}
	return diff;
}

// Put the collected host images into the shadow and the mirror.  No host callback runs in here.
inline bool apply_patches(HnswMetadata *meta, Entry *e)
{
	const size_t esz = meta->size_data_per_element, maxM = meta->maxM;
	Patches &pt = patches();
	const label_t dead = (label_t) 1 << HNSW_LABEL_DELETED_BIT;
	e->suspect = true;      // shadow and mirror change in steps: if this function is left half-way (an allocation fails, an upload
							// fails and the re-mirror after it fails too) the entry is never picked again
	// the mirror must hold every element number the new images name (link targets are checked on upload)
	size_t need = e->n;
	for (size_t i = 0; i < pt.ids.size(); i++)
	{
		need = std::max(need, (size_t) pt.ids[i] + 1);
		const uint32_t *l = reinterpret_cast<const uint32_t *>(pt.images.data() + i * esz);
		const uint32_t cnt = l[0] <= maxM ? l[0] : (uint32_t) maxM;
		for (uint32_t j = 1; j <= cnt; j++) need = std::max(need, (size_t) l[j] + 1);
	}
	if (need > e->n)
	{
		const size_t old = e->n;
		e->shadow.resize(need * esz);
		for (size_t s = old; s < need; s++)                     // placeholders: zero-linked, vacuum-flagged, as host_walk.h leaves holes
		{
			memset(e->shadow.data() + s * esz, 0, esz);
			memcpy(e->shadow.data() + s * esz + meta->offset_label, &dead, sizeof(dead));
		}
		if (hnsw_gpu_index_update_from_flat(e->ix, e->shadow.data() + old * esz, old, need - old) != HNSW_GPU_OK) return false;
		e->n = need;
		if (e->stamp.size() < need) e->stamp.resize(need, 0);
	}
	// images into the shadow, then contiguous runs into the mirror
	std::vector<uint32_t> order(pt.ids.size());
	for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t) i;
	std::sort(order.begin(), order.end(), [&pt](uint32_t a, uint32_t b) { return pt.ids[a] < pt.ids[b]; });
	for (uint32_t i : order) memcpy(e->shadow.data() + (size_t) pt.ids[i] * esz, pt.images.data() + (size_t) i * esz, esz);
	for (size_t i = 0; i < order.size();)
	{
		size_t j = i + 1;
		while (j < order.size() && pt.ids[order[j]] == pt.ids[order[j - 1]] + 1) j++;
		const uint32_t first = pt.ids[order[i]];
		if (hnsw_gpu_index_update_from_flat(e->ix, e->shadow.data() + (size_t) first * esz, first, j - i) != HNSW_GPU_OK) return false;
		i = j;
	}
	stats().patched += pt.ids.size();
	e->suspect = false;
	return true;
}

constexpr size_t POPS_CAP = 1 << 14;       // pops of one walk that can be validated (ef = 128: ~150; scans that double ef: thousands)
inline std::vector<uint32_t> &popbuf() { static thread_local std::vector<uint32_t> b; return b; }

// hnsw_search over the cache.  labels: ef entries.  Returns false on failure (message on stderr by the caller).
inline bool search(HnswMetadata *meta, const coord_t *point, size_t ef, label_t *labels, uint32_t *count, int device)
{
	bool empty = false;
	Entry *e = pick(meta, &empty);
	if (empty) { *count = 0; return true; }
	stats().searches++;
	if (!e)
	{
		e = resnapshot(meta, nullptr, device);
		if (!e) return false;
		const bool ok = hnsw_gpu_search_batch(e->ix, point, 1, ef, labels, nullptr, count) == HNSW_GPU_OK;   // just built from the host
		if (e->ephemeral) drop(e);
		return ok;
	}
	std::vector<uint32_t> &pops = popbuf();
	if (pops.size() < POPS_CAP + 64) pops.resize(POPS_CAP + 64);
	for (int round = 0; round < 12; round++)
	{
		stats().search_rounds++;
		const long diff = traced_walk(meta, e, point, ef, 0, labels, count, pops, POPS_CAP, false);
		if (diff == -2) return false;
		if (diff < 0) break;
		if (diff == 0) return true;
		if ((size_t) diff > 64 + e->n / 4) break;                // mostly another index: re-mirroring is cheaper
		if (!apply_patches(meta, e)) break;
	}
	stats().fallbacks++;
	e = resnapshot(meta, e, device);
	if (!e) return false;
	const bool ok = hnsw_gpu_search_batch(e->ix, point, 1, ef, labels, nullptr, count) == HNSW_GPU_OK;
	if (e->ephemeral) drop(e);
	return ok;
}

// Before hnsw_bind_point(idx) runs on the device: a mirror in which every element the insert will read equals the
// host's.  The insert reads what searchBaseLayer(efConstruction) touches, and — through
// mutuallyConnectNewElement / getNeighborsByHeuristic (hnswalg.cpp:117-222) — the link lists of the selected
// neighbours and the vectors of THEIR link targets; the selected neighbours are among the search's results, so the
// results are expanded like pops.  Returns null on failure.
// What the validation walk of prepare_insert found: searchBaseLayer(point, ef = efConstruction) on the mirror, element numbers and
// distances ascending by (dist, idx) — exactly what the insert's own search would return a moment later (same query, same graph:
// the new row is not linked yet), so the device insert takes it instead of walking again (hnsw_gpu_index_insert_candidates).
// valid only when prepare_insert returned through a clean validation (not after a re-mirror: no walk ran then).
struct InsertWalk { bool valid = false; uint32_t cnt = 0; std::vector<idx_t> idx; std::vector<dist_t> dist; };
inline InsertWalk &insert_walk() { static thread_local InsertWalk w; return w; }

inline Entry *prepare_insert(HnswMetadata *meta, const coord_t *point, idx_t idx, int device)
{
	bool empty = false;
	insert_walk().valid = false;
	Entry *e = pick(meta, &empty);
	stats().inserts++;
	if (empty) return nullptr;                                  // (idx 0 never gets here; an index without entry point cannot take idx > 0)
	if (e && e->n > (size_t) idx) e = nullptr;                  // holds elements the host does not have yet: not this index
	if (!e) return resnapshot(meta, nullptr, device);
	const size_t efc = std::max<size_t>(meta->efConstruction, 1);
	std::vector<uint32_t> &pops = popbuf();
	if (pops.size() < POPS_CAP + efc + 64) pops.resize(POPS_CAP + efc + 64);
	static thread_local std::vector<label_t> res;
	res.resize(efc);
	InsertWalk &iw = insert_walk();
	iw.dist.resize(efc);
	iw.idx.resize(efc);
	for (int round = 0; round < 12; round++)
	{
		uint32_t cnt = 0;
		stats().insert_rounds++;
		const long diff = traced_walk(meta, e, point, efc, 1, res.data(), &cnt, pops, POPS_CAP, true, iw.dist.data());
		if (diff == -2) return nullptr;
		if (diff < 0) break;
		if (diff == 0)
		{
			for (uint32_t i = 0; i < cnt; i++) iw.idx[i] = (idx_t) res[i];
			iw.cnt = cnt;
			iw.valid = true;
			return e;
		}
		if ((size_t) diff > 64 + e->n / 4) break;
		if (!apply_patches(meta, e)) break;
	}
	stats().fallbacks++;
	return resnapshot(meta, e, device);
}

// The shadow follows what the insert did to the mirror (and, through the write-back, to the host).
inline void shadow_set_links(HnswMetadata *meta, Entry *e, idx_t idx, const idx_t *list /* count + maxM links */)
{
	if ((size_t) idx < e->n) memcpy(e->shadow.data() + (size_t) idx * meta->size_data_per_element, list, (meta->maxM + 1) * sizeof(idx_t));
}

inline bool shadow_append(HnswMetadata *meta, Entry *e, size_t upto /* new element count */, idx_t idx, const coord_t *point, label_t label)
{
	const size_t esz = meta->size_data_per_element;
	const label_t dead = (label_t) 1 << HNSW_LABEL_DELETED_BIT;
	try { e->shadow.resize(upto * esz); if (e->stamp.size() < upto) e->stamp.resize(upto, 0); } catch (...) { return false; }
	for (size_t s = e->n; s < upto; s++)
	{
		memset(e->shadow.data() + s * esz, 0, esz);
		memcpy(e->shadow.data() + s * esz + meta->offset_label, &dead, sizeof(dead));
	}
	char *img = e->shadow.data() + (size_t) idx * esz;
	memset(img, 0, esz);
	memcpy(img + meta->offset_data, point, meta->dim * sizeof(coord_t));
	memcpy(img + meta->offset_label, &label, sizeof(label));
	e->n = upto;
	return true;
}

}  // namespace shimcache
