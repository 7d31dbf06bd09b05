"""Multi-GPU layouts of the search path — one process per GPU.

* :func:`query_slice` — replicas + query sharding: every rank mirrors the whole index and takes a
  contiguous slice of the query batch; no data-path collective (used by ``bench.py --gpus N``).
* :class:`ShardedIndex` — row-sharded index for indexes that are split across GPUs (config C4):
  contiguous row ranges, an independent graph per shard (entry = shard-local element 0), every
  rank searches every query on its shard, then ONE exchange: an all-gather of the per-shard
  (dist, label) lists over RCCL (``torch.distributed`` backend "nccl") followed by the device
  merge kernel.  Labels carry the global row number, so they are unique across shards.  The exchange is
  latency-bound (nq*ef*12 bytes per rank), hence ONE all-gather of one packed block per rank
  ([labels | dists]) and no chunking; the device merge (``hnsw_gpu_merge_topk_strided_dev``) reads the lists where the
  all-gather put them.  Nothing in the step allocates or packs: the local search writes labels and distances STRAIGHT into
  this rank's packed block; the block and the gather buffer are allocated once per (batch size, beam) and reused, the merge
  writes into the caller's tensors when given (``search(..., out=)``).  (Rounds 1-5 packed through three extra kernels and two
  allocations per search — :func:`pack_block`, kept for callers that bring their own result tensors.)

The same layout inside one process (several devices, no torch.distributed) is native:
``hnsw_gpu_sharded_*`` in include/hnsw_gpu.h / :class:`pg_embedding_amd.LocalShardedIndex`.

The collective and the control flow are device-agnostic torch code, so the N>1 path is covered on
CPU by world-size-2 gloo tests with the local search / merge steps injected by the test.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range [lo, hi) of `rank`."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def block_bytes(nq: int, ef: int) -> int:
    """Bytes of one rank's packed result block: nq*ef labels (8 B) then nq*ef distances (4 B), padded to 16."""
    return (nq * ef * 12 + 15) // 16 * 16


def pack_block(labels, dists):
    """[nq, ef] int64 labels + [nq, ef] float32 dists -> one uint8 block (what ONE all-gather moves)."""
    import torch
    nq, ef = labels.shape
    blk = torch.zeros(block_bytes(nq, ef), dtype=torch.uint8, device=labels.device)
    blk[:nq * ef * 8] = labels.contiguous().view(torch.uint8).reshape(-1)
    blk[nq * ef * 8:nq * ef * 12] = dists.contiguous().view(torch.uint8).reshape(-1)
    return blk


def unpack_blocks(blocks, nq: int, ef: int):
    """[world, block] uint8 -> (labels[world, nq, ef] int64, dists[world, nq, ef] float32) (copies)."""
    import torch
    world = blocks.shape[0]
    lab = blocks[:, :nq * ef * 8].contiguous().view(torch.int64).reshape(world, nq, ef)
    dst = blocks[:, nq * ef * 8:nq * ef * 12].contiguous().view(torch.float32).reshape(world, nq, ef)
    return lab, dst


def query_slice(n_queries: int, world: int, rank: int) -> Tuple[int, int]:
    """Query range of `rank` when the index is replicated."""
    return n_queries * rank // world, n_queries * (rank + 1) // world


class ShardedIndex:
    """One shard of a row-partitioned index + the exchange step.

    local_search(queries, ef) -> (labels[nq, ef] int64, dists[nq, ef] float32) ascending by
        (dist, label), unused tail = (-1 / all-ones, +inf)   [default: the device search]
    merge(labels[world, nq, ef], dists[world, nq, ef], ef) -> (labels[nq, ef], dists[nq, ef],
        counts[nq])                      [injected by the CPU tests; default: the device merge kernel
                                          reading the gathered blocks in place, merge_packed_torch]
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

            def local_search(q, ef, out=None):
                res = index.search_torch(q, ef, out=out)
                return res["labels"], res["dists"]
            local_search.writes_in_place = True          # (takes `out`: labels / dists / counts tensors to write into)
        self.merge_packed = None
        if merge is None:
            from .index import merge_packed_torch
            self.merge_packed = merge_packed_torch
        self.local_search = local_search
        self.merge = merge
        self._bufs = {}                        # (nq, ef, device) -> this rank's packed block (+ views into it) and the gather buffer
        self.exchanges = 0                     # collectives issued so far (one per search)
        self.record_timing = False             # bench.py --mode sharded: stamp the three steps of every search
        self._stamps = []                      # per search: 4 device events (or 4 host times for CPU tensors)

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

    def search(self, queries, ef: int, out=None):
        """Every rank passes the SAME queries; every rank returns the merged result (labels[nq, ef], dists[nq, ef], counts[nq]).
        out: those three tensors to write into (a caller that searches in a loop allocates them once); allocated per call otherwise."""
        import torch

        def stamp():
            if not self.record_timing:
                return None
            if queries.is_cuda:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(queries.device))
                return e
            import time
            return time.perf_counter()

        t0 = stamp()
        nq = queries.shape[0]
        b = self._buffers(nq, ef, queries.device)
        if getattr(self.local_search, "writes_in_place", False):
            # labels | dists land in this rank's packed block as the search kernel writes them: no pack step at all
            self.local_search(queries, ef, out={"labels": b["labels"], "dists": b["dists"], "counts": b["counts"]})
        else:                                   # an injected search (the CPU tests) returns its own tensors: one copy each
            labels, dists = self.local_search(queries, ef)
            b["labels"].copy_(labels)
            b["dists"].copy_(dists)
        t1 = stamp()
        mine = b["mine"]
        if self.world == 1:
            blocks = mine.unsqueeze(0)
        else:
            flat = b["flat"]
            blocks = flat.view(self.world, mine.numel())
            # THE exchange step: one all-gather.  (gloo with device tensors — the 2-process test on a
            # 1-GPU box — only has the list form; over RCCL and for CPU tensors the flat form is used.)
            if mine.is_cuda and self.dist.get_backend(self.group) != "nccl":
                self.dist.all_gather(list(blocks.unbind(0)), mine, group=self.group)
            else:
                self.dist.all_gather_into_tensor(flat, mine, group=self.group)
            self.exchanges += 1
        t2 = stamp()
        if self.merge_packed is not None:
            res = self.merge_packed(blocks, nq, ef, out=out)
        else:
            lab, dst = unpack_blocks(blocks, nq, ef)
            res = self.merge(lab, dst, ef)
        if self.record_timing:
            self._stamps.append((t0, t1, t2, stamp()))
        return res

    def _buffers(self, nq: int, ef: int, device):
        """the exchange step's memory for one (batch size, beam): allocated at the first search of that shape, reused ever after"""
        import torch
        key = (nq, ef, str(device))
        b = self._bufs.get(key)
        if b is None:
            nb = block_bytes(nq, ef)
            mine = torch.zeros(nb, dtype=torch.uint8, device=device)
            b = {"mine": mine,
                 "labels": mine[:nq * ef * 8].view(torch.int64).reshape(nq, ef),
                 "dists": mine[nq * ef * 8:nq * ef * 12].view(torch.float32).reshape(nq, ef),
                 "counts": torch.empty(nq, dtype=torch.int32, device=device),
                 "flat": torch.empty(self.world * nb, dtype=torch.uint8, device=device) if self.world > 1 else None}
            self._bufs[key] = b
        return b

    def timings_ms(self):
        """[(local_search_ms, exchange_ms, merge_ms)] of the searches since record_timing was switched on (device
        events on the search stream; call after a synchronize)."""
        out = []
        for t0, t1, t2, t3 in self._stamps:
            if isinstance(t0, float):
                out.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            else:
                out.append((t0.elapsed_time(t1), t1.elapsed_time(t2), t2.elapsed_time(t3)))
        return out
