"""Host-side mirror of the index scan that calls the hot path (embedding.c:284-370,
`hnsw_gettuple`): hand out the results of one `hnsw_search`; when they are exhausted and the
search was "full" (n_results == efSearch) double efSearch, search again and continue with the
rows not returned yet.

The reference cannot see distances (hnsw_search returns labels only), so it de-duplicates the
second result list against the first by sorting TIDs (embedding.c:345-363) and appends what is
new in the new list's order.  `IndexScan` reproduces exactly that sequence.
"""
from __future__ import annotations

from typing import Iterator, Optional

import numpy as np

from .index import GpuIndex


class IndexScan:
    """``for label in IndexScan(index, query): ...`` == repeated amgettuple calls."""

    def __init__(self, index: GpuIndex, query, efsearch: Optional[int] = None, max_ef: Optional[int] = None):
        self.index = index
        self.query = np.ascontiguousarray(query, dtype=np.float32).reshape(1, -1)
        if self.query.shape[1] != index.meta.dim:       # embedding.c:311-315
            raise ValueError(f"Wrong number of dimensions: {self.query.shape[1]} instead of "
                             f"{index.meta.dim} expected")
        self.ef = int(efsearch or index.meta.efSearch)
        self.max_ef = max_ef
        self.results = None          # labels handed out or pending, in hand-out order
        self.curr = 0
        self.no_more = False

    def _search(self):
        labels, _, counts = self.index.search(self.query, self.ef)
        return labels[0, :int(counts[0])]

    def __iter__(self) -> Iterator[int]:
        return self

    def __next__(self) -> int:
        if self.results is None:                                   # first call, embedding.c:296-328
            r = self._search()
            self.results = list(r.tolist())
            self.no_more = len(r) < self.ef                        # :322
        if self.curr >= len(self.results):                         # :329
            # (the reference has no cap: the doubling ends when a search comes back short; `max_ef` is an
            # optional safety valve for callers, the device path itself clamps the beam to the index size)
            if self.no_more or (self.max_ef is not None and self.ef * 2 > self.max_ef):
                raise StopIteration
            self.ef *= 2                                           # :334
            r = self._search()
            if len(r) <= len(self.results):                        # :338-342 no new results found
                raise StopIteration
            self.no_more = len(r) < self.ef                        # :343
            seen = set(self.results)                               # qsort + bsearch, :355-363
            self.results.extend(int(x) for x in r.tolist() if int(x) not in seen)
            if self.curr >= len(self.results):
                raise StopIteration
        v = self.results[self.curr]
        self.curr += 1
        return v
