"""Row-sharded search across the GPUs of one node: one process per GPU, each holding a contiguous
block of rows; every rank searches the same query batch on its block, the per-rank top-k lists are
exchanged with ONE all-gather (RCCL over xGMI when the process group is `nccl`) and merged by
(distance, global id).

Mirrors the reference's in-process sharding: MultiGpuShardManager::search
(src/experimental/gpu/mod.rs:1355-1366: scan every shard, concat, sort, truncate) and
merge_shard_results (src/persistence/shard.rs:622-643).  Rows are placed in contiguous blocks
(global id = block start + local id) rather than round-robin, so the tie-break by id survives the merge.

The local search and the merge are injected callables: the product wiring passes the HIP entry points;
nothing here computes distances.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by `rank`: blocks of ceil(n/world) rows."""
    per = -(-n_total // world) if world > 0 else n_total
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


class ShardedSearcher:
    """search(queries) -> (ids[q,k] int64 global, dists[q,k] f32, counts[q] int32) on every rank.

    local_search(queries, k) -> (ids[q,k] int64 LOCAL, dists[q,k], counts[q] int32), padded rows hold id -1.
    merge(all_dists[w,q,k], all_ids[w,q,k], all_counts[w,q], k) -> (ids, dists, counts).
    """

    def __init__(self, local_search: Callable, merge: Callable, row_offset: int, group=None):
        self.local_search = local_search
        self.merge = merge
        self.row_offset = int(row_offset)
        self.group = group

    def search(self, queries: torch.Tensor, k: int):
        ids, dists, counts = self.local_search(queries, k)
        ids = torch.where(ids >= 0, ids + self.row_offset, ids)  # local -> global, padding stays -1
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return ids, dists, counts
        q = ids.shape[0]
        all_ids = torch.empty((world, q, k), dtype=ids.dtype, device=ids.device)
        all_dists = torch.empty((world, q, k), dtype=dists.dtype, device=dists.device)
        all_counts = torch.empty((world, q), dtype=counts.dtype, device=counts.device)
        # three small collectives (8*q*k + 4*q*k + 4*q bytes per rank): latency-bound, not link-bound
        dist.all_gather_into_tensor(all_ids.view(-1), ids.contiguous().view(-1), group=self.group)
        dist.all_gather_into_tensor(all_dists.view(-1), dists.contiguous().view(-1), group=self.group)
        dist.all_gather_into_tensor(all_counts.view(-1), counts.contiguous().view(-1), group=self.group)
        return self.merge(all_dists, all_ids, all_counts, k)


def hip_local_search(index, precision: str = "fp32") -> Callable:
    """local_search bound to a resident needle_amd.FlatIndex (device tensors in and out)."""

    def run(queries: torch.Tensor, k: int):
        q = queries.shape[0]
        ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=queries.device)
        dists = torch.empty((q, max(k, 1)), dtype=torch.float32, device=queries.device)
        counts = torch.empty((q,), dtype=torch.int32, device=queries.device)
        torch.cuda.current_stream(queries.device).synchronize()  # the library runs on its own stream
        index.search_device(queries, k, ids, dists, counts, precision=precision)
        return ids[:, :k], dists[:, :k], counts  # padded ids are UINT64_MAX == -1 as int64

    return run


def hip_local_search_handle(handle, nprobe: int = 0, precision: str = "fp32") -> Callable:
    """local_search bound to any resident index handle (IVF, SQ8, PQ, binary): a shard holds its row block's slice of
    every inverted list / of the code matrix, with centroids, codebooks and quantizer parameters replicated, so the
    probe lists are the same on every rank and the union of the shards' candidates is the whole index's."""
    import ctypes as C

    from . import _lib
    from .index import PRECISION

    def run(queries: torch.Tensor, k: int):
        q = queries.shape[0]
        ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=queries.device)
        dists = torch.empty((q, max(k, 1)), dtype=torch.float32, device=queries.device)
        counts = torch.empty((q,), dtype=torch.int32, device=queries.device)
        torch.cuda.current_stream(queries.device).synchronize()
        sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0, filter_bitmask=None)
        _lib.check(_lib.lib().needle_hip_batch_search_device(handle, queries.data_ptr(), q, C.byref(sp), ids.data_ptr(),
                                                             dists.data_ptr(), counts.data_ptr()))
        return ids[:, :k], dists[:, :k], counts

    return run


def hip_merge(ctx) -> Callable:
    """merge bound to needle_hip_merge_topk_device."""

    def run(all_dists: torch.Tensor, all_ids: torch.Tensor, all_counts: torch.Tensor, k: int):
        w, q, per = all_ids.shape
        out_ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=all_ids.device)
        out_d = torch.empty((q, max(k, 1)), dtype=torch.float32, device=all_ids.device)
        out_c = torch.empty((q,), dtype=torch.int32, device=all_ids.device)
        torch.cuda.current_stream(all_ids.device).synchronize()
        ctx.merge_topk_device(all_dists.contiguous(), all_ids.contiguous(), all_counts.contiguous(), w, q, per, k,
                              out_ids, out_d, out_c)
        return out_ids[:, :k], out_d[:, :k], out_c

    return run
