"""Multi-GPU layouts of the search path — one process per GPU.

* :func:`query_slice` — replicas + query sharding: every rank mirrors the whole index and takes a
  contiguous slice of the query batch; no data-path collective (used by ``bench.py --gpus N``).
* :class:`ShardedIndex` — row-sharded index for indexes that are split across GPUs (config C4):
  contiguous row ranges, an independent graph per shard (entry = shard-local element 0), every
  rank searches every query on its shard, then ONE exchange: an all-gather of the per-shard
  (dist, label) lists over RCCL (``torch.distributed`` backend "nccl") followed by the device
  merge kernel (``hnsw_gpu_merge_topk_dev``).  Labels carry the global row number, so they are
  unique across shards.  The exchange is latency-bound (nq*ef*12 bytes per rank), hence a single
  all-gather and no chunking.

The collective and the control flow are device-agnostic torch code, so the N>1 path is covered on
CPU by world-size-2 gloo tests with the local search / merge steps injected by the test.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range [lo, hi) of `rank`."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def query_slice(n_queries: int, world: int, rank: int) -> Tuple[int, int]:
    """Query range of `rank` when the index is replicated."""
    return n_queries * rank // world, n_queries * (rank + 1) // world


class ShardedIndex:
    """One shard of a row-partitioned index + the exchange step.

    local_search(queries, ef) -> (labels[nq, ef] int64, dists[nq, ef] float32) ascending by
        (dist, label), unused tail = (-1 / all-ones, +inf)   [default: the device search]
    merge(labels[world, nq, ef], dists[world, nq, ef], ef) -> (labels[nq, ef], dists[nq, ef],
        counts[nq])                                          [default: the device merge kernel]
    """

    def __init__(self, index=None, local_search: Optional[Callable] = None,
                 merge: Optional[Callable] = None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.index = index
        if local_search is None:
            if index is None:
                raise ValueError("need a GpuIndex or a local_search callable")

            def local_search(q, ef):
                out = index.search_torch(q, ef)
                return out["labels"], out["dists"]
        if merge is None:
            from .index import merge_topk_torch
            merge = merge_topk_torch
        self.local_search = local_search
        self.merge = merge

    @classmethod
    def build(cls, rows, global_first: int, meta, device: int = 0, max_batch: int = 0, ratio: int = 0):
        """Build this rank's shard on its GPU from device rows; label = global row number."""
        import torch
        from .index import GpuIndex
        n = rows.shape[0]
        ix = GpuIndex.empty(meta, n, device=device)
        labels = torch.arange(global_first, global_first + n, dtype=torch.int64, device=rows.device)
        ix.append_torch(rows.contiguous(), labels)
        ix.link(0, n, max_batch, ratio, torch.cuda.current_stream(rows.device).cuda_stream)
        return cls(index=ix)

    def search(self, queries, ef: int):
        """Every rank passes the SAME queries; every rank returns the merged result."""
        import torch
        labels, dists = self.local_search(queries, ef)
        if self.world == 1:
            all_l, all_d = labels.unsqueeze(0), dists.unsqueeze(0)
        else:
            all_l = torch.empty((self.world,) + tuple(labels.shape), dtype=labels.dtype, device=labels.device)
            all_d = torch.empty((self.world,) + tuple(dists.shape), dtype=dists.dtype, device=dists.device)
            # one all-gather per array into views of the [world, nq, ef] buffers
            self.dist.all_gather(list(all_l.unbind(0)), labels.contiguous(), group=self.group)
            self.dist.all_gather(list(all_d.unbind(0)), dists.contiguous(), group=self.group)
        return self.merge(all_l.contiguous(), all_d.contiguous(), ef)
