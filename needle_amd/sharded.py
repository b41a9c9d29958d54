"""Row-sharded search across the GPUs of one node: one process per GPU, each holding a contiguous
block of rows; every rank searches the same query batch on its block, the per-rank top-k lists are
exchanged with ONE all-gather and merged by (distance, global id) on every rank.

Mirrors the reference's in-process sharding: MultiGpuShardManager::search
(src/experimental/gpu/mod.rs:1355-1366: scan every shard, concat, sort, truncate) and
merge_shard_results (src/persistence/shard.rs:622-643).  Rows are placed in contiguous blocks
(global id = block start + local id) rather than round-robin, so the tie-break by id survives the merge.

Two transports:
  * `RcclShardedSearcher` -- the product path: needle_hip_sharded_batch_search_device does local search,
    id globalisation, the RCCL all-gather over xGMI and the merge on the library's stream, behind the C ABI.
  * `ShardedSearcher` -- the same steps with the exchange over a `torch.distributed` group (gloo rehearsals
    of the multi-rank control flow on a box with fewer GPUs than ranks, and CPU tests): the local search and
    the merge are injected callables; nothing here computes distances.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Rows [lo, hi) owned by `rank`: blocks of ceil(n/world) rows, rounded up to a multiple of `align`."""
    per = -(-n_total // world) if world > 0 else n_total
    per = -(-per // align) * align
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def slot_bytes(q: int, k: int) -> int:
    """Bytes of one rank's slot of the exchange buffer: ids [q][k] u64 | distances [q][k] f32 | counts [q] u32,
    rounded up to 16 (needle_hip_shard_slot_bytes)."""
    return (q * k * 12 + q * 4 + 15) // 16 * 16


def pack_slot(slot: torch.Tensor, ids: torch.Tensor, dists: torch.Tensor, counts: torch.Tensor) -> None:
    """Write one rank's (ids int64 [q,k], dists f32 [q,k], counts int32 [q]) into its uint8 slot."""
    q, k = ids.shape
    slot[: q * k * 8].view(torch.int64).copy_(ids.reshape(-1))
    slot[q * k * 8: q * k * 12].view(torch.float32).copy_(dists.reshape(-1))
    slot[q * k * 12: q * k * 12 + q * 4].view(torch.int32).copy_(counts.reshape(-1))


def unpack_slots(slots: torch.Tensor, world: int, q: int, k: int):
    """[world, slot_bytes] uint8 -> (ids [world,q,k] int64, dists [world,q,k] f32, counts [world,q] int32) copies."""
    s = slots.view(world, -1)
    ids = torch.stack([s[w, : q * k * 8].view(torch.int64).view(q, k) for w in range(world)])
    dists = torch.stack([s[w, q * k * 8: q * k * 12].view(torch.float32).view(q, k) for w in range(world)])
    counts = torch.stack([s[w, q * k * 12: q * k * 12 + q * 4].view(torch.int32) for w in range(world)])
    return ids, dists, counts


class ShardedSearcher:
    """search(queries, k) -> (ids[q,k] int64 global, dists[q,k] f32, counts[q] int32) on every rank.

    local_search(queries, k) -> (ids[q,k] int64 LOCAL, dists[q,k], counts[q] int32), padded rows hold id -1.
    merge(slots[world, slot_bytes] uint8, world, q, k) -> (ids, dists, counts).
    The exchange is ONE all_gather_into_tensor of the packed slots over `group` (host tensors for gloo).
    """

    def __init__(self, local_search: Callable, merge: Callable, row_offset: int, group=None):
        self.local_search = local_search
        self.merge = merge
        self.row_offset = int(row_offset)
        self.group = group

    def search(self, queries, k: int):
        ids, dists, counts = self.local_search(queries, k)
        ids = torch.where(ids >= 0, ids + self.row_offset, ids)  # local -> global, padding stays -1
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1 or k == 0:
            return ids, dists, counts
        q = ids.shape[0]
        rank = dist.get_rank(self.group)
        nbytes = slot_bytes(q, k)
        on_host = dist.get_backend(self.group) == "gloo"
        buf_dev = torch.device("cpu") if on_host else ids.device
        slots = torch.zeros((world, nbytes), dtype=torch.uint8, device=buf_dev)
        pack_slot(slots[rank], ids.to(buf_dev), dists.to(buf_dev), counts.to(buf_dev))
        dist.all_gather_into_tensor(slots.view(-1), slots[rank].clone(), group=self.group)  # the one collective
        return self.merge(slots.to(ids.device), world, q, k)


class Communicator:
    """needle_hip_comm: one RCCL communicator over the ranks of a torch.distributed group (the unique id travels over
    that group, which is the host's own channel between its processes)."""

    def __init__(self, ctx, group=None):
        from . import _lib

        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        if rank == 0:
            _lib.check(_lib.lib().needle_hip_comm_unique_id(uid))
        box = [bytes(uid)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        self.ctx, self.rank, self.world = ctx, rank, world
        _lib.check(_lib.lib().needle_hip_comm_init(ctx._h, uid, rank, world, C.byref(self._h)))

    def close(self) -> None:
        from . import _lib

        from .index import _destroy_handle

        _destroy_handle(self, "needle_hip_comm_destroy")


class RcclShardedSearcher:
    """The product path: one C-ABI call per batch (needle_hip_sharded_batch_search_device)."""

    def __init__(self, handle, comm: Communicator, row_offset: int, nprobe: int = 0, precision: str = "bf16"):
        self.handle, self.comm, self.row_offset, self.nprobe, self.precision = handle, comm, int(row_offset), nprobe, precision

    def search(self, queries: torch.Tensor, k: int):
        from . import _lib
        from .index import PRECISION, _wait_for_producer

        q = queries.shape[0]
        ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=queries.device)
        dists = torch.empty((q, max(k, 1)), dtype=torch.float32, device=queries.device)
        counts = torch.empty((q,), dtype=torch.int32, device=queries.device)
        _wait_for_producer(queries)  # the library runs on its own streams: the queries must be complete
        sp = _lib.SearchParams(k=k, nprobe=self.nprobe, precision=PRECISION[self.precision], reserved=0, filter_bitmask=None)
        _lib.check(_lib.lib().needle_hip_sharded_batch_search_device(self.handle, self.comm._h, self.row_offset,
                                                                     queries.data_ptr(), q, C.byref(sp), ids.data_ptr(),
                                                                     dists.data_ptr(), counts.data_ptr()))
        return ids[:, :k], dists[:, :k], counts


def hip_local_search(index, precision: str = "fp32") -> Callable:
    """local_search bound to a resident needle_amd.FlatIndex (device tensors in and out)."""
    return hip_local_search_handle(index._h, 0, precision)


def hip_local_search_handle(handle, nprobe: int = 0, precision: str = "fp32") -> Callable:
    """local_search bound to any resident index handle (flat, IVF, SQ8, PQ, binary): a shard holds its row block's slice of
    every inverted list / of the code matrix, with centroids, codebooks and quantizer parameters replicated, so the
    probe lists are the same on every rank and the union of the shards' candidates is the whole index's."""
    from . import _lib
    from .index import PRECISION, _wait_for_producer

    def run(queries: torch.Tensor, k: int):
        q = queries.shape[0]
        ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=queries.device)
        dists = torch.empty((q, max(k, 1)), dtype=torch.float32, device=queries.device)
        counts = torch.empty((q,), dtype=torch.int32, device=queries.device)
        _wait_for_producer(queries)  # the library runs on its own stream
        sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0, filter_bitmask=None)
        _lib.check(_lib.lib().needle_hip_batch_search_device(handle, queries.data_ptr(), q, C.byref(sp), ids.data_ptr(),
                                                             dists.data_ptr(), counts.data_ptr()))
        return ids[:, :k], dists[:, :k], counts  # padded ids are UINT64_MAX == -1 as int64

    return run


def hip_merge(ctx) -> Callable:
    """merge bound to needle_hip_merge_shard_slots_device (the merge kernel of the RCCL path, on gathered slots)."""
    from . import _lib

    def run(slots: torch.Tensor, world: int, q: int, k: int):
        out_ids = torch.empty((q, max(k, 1)), dtype=torch.int64, device=slots.device)
        out_d = torch.empty((q, max(k, 1)), dtype=torch.float32, device=slots.device)
        out_c = torch.empty((q,), dtype=torch.int32, device=slots.device)
        torch.cuda.current_stream(slots.device).synchronize()
        _lib.check(_lib.lib().needle_hip_merge_shard_slots_device(ctx._h, slots.data_ptr(), world, q, k, out_ids.data_ptr(),
                                                                  out_d.data_ptr(), out_c.data_ptr()))
        return out_ids[:, :k], out_d[:, :k], out_c

    return run
