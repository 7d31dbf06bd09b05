// host_walk.h — copy a host index into one contiguous array of element images using nothing but the
// accessor the reference's search itself uses: hnsw_begin_read / hnsw_end_read (embedding.c:704-767).
// Shared by libembedding_gpu.so (hnsw_gpu_shim_snapshot) and libembedding_gpuc.so (upload to the server).
//
// The walk FOLLOWS THE LINKS from the entry point (element 0, hnswalg.cpp:55) instead of counting
// element numbers up until one is missing: the callback interface has no "how many elements" call, and
// in the real host a probe past the last page is not a polite `false` — ReadBuffer raises
// ERROR "could not read block" and the transaction is gone (embedding.c:728).  Links only ever name
// elements that exist, so a link-following walk cannot step outside the relation.  It is also exact:
// an element that cannot be reached from the entry point cannot be visited by searchBaseLayer, so it
// can neither be returned by a search nor be linked to by an insert — leaving it out changes no answer.
// Element numbers that were not reached (unreachable elements, and the tail-of-page holes of
// embedding.c:229,693) become zero-linked, vacuum-flagged placeholders.  One pin at a time (the host
// allows 4, embedding.c:40), nothing allocated across a callback that is not reclaimed on the next
// walk (a callback may leave by longjmp: elog(ERROR)).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "hnsw_abi.h"

namespace hostwalk {

// grow(bytes) must return the base of a buffer of at least `bytes` bytes whose old contents are kept
// (or nullptr).  Returns the number of element slots in the image (highest reached number + 1),
// 0 for an empty index, -1 when the buffer cannot grow.
template <class Grow>
inline long copy_reachable(HnswMetadata *meta, Grow grow)
{
	static thread_local std::vector<uint32_t> todo;
	static thread_local std::vector<uint64_t> seen;
	todo.clear();
	seen.clear();
	const size_t esz = meta->size_data_per_element, maxM = meta->maxM;
	const label_t dead = (label_t) 1 << HNSW_LABEL_DELETED_BIT;
	char *base = nullptr;
	size_t slots = 0;                     // element slots initialised so far
	size_t n = 0;
	// every element number is pushed at most once (marked when pushed)
	auto mark = [](uint32_t i) -> bool {
		if ((size_t) i / 64 >= seen.size()) seen.resize((size_t) i / 64 + 1 + seen.size() / 2, 0);
		const uint64_t bit = 1ull << (i % 64);
		if (seen[i / 64] & bit) return false;
		seen[i / 64] |= bit;
		return true;
	};
	mark(meta->enterpoint_node);
	todo.push_back(meta->enterpoint_node);
	while (!todo.empty())
	{
		const uint32_t idx = todo.back();
		todo.pop_back();
		idx_t *links = nullptr;
		if (!hnsw_begin_read(meta, (idx_t) idx, &links, nullptr, nullptr))
		{
			if (idx == meta->enterpoint_node) return 0;        // empty index (hnswalg.cpp:56-57)
			continue;                                          // a dangling link: the reference skips it too
		}
		if ((size_t) idx >= slots)
		{
			// (doubling while small, then in steps of a quarter: the image of a large index must not cost twice its size)
			size_t want = slots ? slots : 256;
			while (want <= (size_t) idx) want += want < (1u << 16) ? want : want / 4;
			base = grow(want * esz);
			if (!base) { hnsw_end_read(meta); return -1; }
			for (size_t s = slots; s < want; s++)
			{
				memset(base + s * esz, 0, esz);
				memcpy(base + s * esz + meta->offset_label, &dead, sizeof(dead));
			}
			slots = want;
		}
		memcpy(base + (size_t) idx * esz, links, esz);         // the element image is contiguous
		hnsw_end_read(meta);
		if ((size_t) idx + 1 > n) n = (size_t) idx + 1;
		const uint32_t *mine = reinterpret_cast<const uint32_t *>(base + (size_t) idx * esz);
		const uint32_t cnt = mine[0] <= maxM ? mine[0] : (uint32_t) maxM;
		for (uint32_t j = 1; j <= cnt; j++)
			if (mark(mine[j])) todo.push_back(mine[j]);
	}
	return (long) n;
}

}  // namespace hostwalk
