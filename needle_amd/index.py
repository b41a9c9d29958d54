"""Thin object layer over the C ABI: a device context and a resident flat index.

Host arrays are numpy; device arrays are anything exposing ``data_ptr()`` (torch
tensors).  PyTorch is plumbing here (device memory and streams for callers that
already hold their data on the GPU); all compute is inside libneedle_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .distance import DistanceFunction

PRECISION = {"fp32": _lib.PRECISION_FP32, "bf16": _lib.PRECISION_BF16, "exact": _lib.PRECISION_EXACT_SCAN}


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def pack_bitmask(bits) -> np.ndarray:
    """bool[n] -> uint32[ceil(n/32)]; bit i lives in word i>>5 at position i&31."""
    b = np.asarray(bits, dtype=np.bool_).reshape(-1)
    pad = (-b.size) % 32
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=np.bool_)])
    return np.packbits(b.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy()


def _mask_words(mask, n):
    if mask is None:
        return None
    m = np.asarray(mask)
    if m.dtype == np.bool_:
        if m.size != n:
            raise ValueError(f"mask has {m.size} entries for {n} rows")
        m = pack_bitmask(m)
    m = np.ascontiguousarray(m, dtype=np.uint32)
    if m.size < (n + 31) // 32:
        raise ValueError("bitmask too short")
    return m


def _wait_for_producer(t) -> None:
    """The library works on its own stream: whatever the caller's stream still has in flight for a device tensor must
    have finished before the library reads it."""
    if getattr(t, "is_cuda", False):
        import torch

        torch.cuda.current_stream(t.device).synchronize()


def _destroy_handle(owner, destroy_name: str = "needle_hip_index_destroy") -> None:
    """Destroys owner._h unless the context it is bound to has been closed: closing a context frees everything bound to
    it, and a wrapper that outlives its context (garbage collected later) must not hand the dead handle to the library
    again."""
    h, ctx = getattr(owner, "_h", None), getattr(owner, "ctx", None)
    ctx_alive = ctx is not None and getattr(ctx, "_h", None) is not None and bool(ctx._h.value)
    if h is not None and h.value and ctx_alive and not _lib.shutting_down():
        getattr(_lib.lib(), destroy_name)(h)
    owner._h = C.c_void_p()


class Context:
    """One HIP device (needle_hip_ctx)."""

    _close_order = 1  # at interpreter exit contexts close after the indexes bound to them

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().needle_hip_init(device, C.byref(self._h)))
        self.device = device
        _lib.register_live(self)

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value and not _lib.shutting_down():
            _lib.lib().needle_hip_shutdown(self._h)
        self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, on: bool) -> None:
        _lib.check(_lib.lib().needle_hip_set_profiling(self._h, int(on)))

    def metrics(self) -> dict:
        m = _lib.Metrics()
        _lib.check(_lib.lib().needle_hip_get_metrics(self._h, C.byref(m)))
        return {name: getattr(m, name) for name, _ in _lib.Metrics._fields_}

    def reset_metrics(self) -> None:
        _lib.check(_lib.lib().needle_hip_reset_metrics(self._h))

    def set_tuning(self, **overrides) -> None:
        """Planning overrides (needle_hip_set_tuning): tests force a code path with them, the A/B tools sweep them.
        `stages=(10, 25)` sets the staged-launch schedule; `reset=1` restores the built-in choices."""
        for key, value in overrides.items():
            if key == "stages":
                vals = list(value) + [0] * (4 - len(value))
                for i, pc in enumerate(vals[:4]):
                    _lib.check(_lib.lib().needle_hip_set_tuning(self._h, f"stage{i}".encode(), float(pc)))
            else:
                _lib.check(_lib.lib().needle_hip_set_tuning(self._h, key.encode(), float(value)))

    def apply_tuning_spec(self, spec: str | None) -> None:
        """`key=value,key=value` as the bench tools take it (stages as `stages=10:25`)."""
        for item in filter(None, (spec or "").split(",")):
            key, _, val = item.partition("=")
            if key == "stages":
                self.set_tuning(stages=[float(x) for x in val.split(":") if x])
            else:
                self.set_tuning(**{key: float(val)})

    # ---- operator-level calls ------------------------------------------
    def batch_distance(self, metric: DistanceFunction, query, base, arith: int = _lib.ARITH_ACCEL_CHUNK4) -> np.ndarray:
        q, b = _f32(query), _f32(base)
        if b.size == 0:
            return np.empty(0, dtype=np.float32)
        if b.ndim != 2 or b.shape[1] != q.size:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "Dimension mismatch")
        out = np.empty(b.shape[0], dtype=np.float32)
        _lib.check(_lib.lib().needle_hip_batch_distance(self._h, int(metric), arith, q.ctypes.data, b.ctypes.data,
                                                        b.shape[0], q.size, out.ctypes.data))
        return out

    def kmeans_assign(self, vectors, centroids, arith: int = _lib.ARITH_ACCEL_CHUNK4) -> np.ndarray:
        v, c = _f32(vectors), _f32(centroids)
        out = np.zeros(v.shape[0], dtype=np.uint32)
        if v.shape[0]:
            _lib.check(_lib.lib().needle_hip_kmeans_assign(self._h, arith, v.ctypes.data, v.shape[0], c.ctypes.data,
                                                           c.shape[0], v.shape[1], out.ctypes.data))
        return out

    def normalize(self, vectors, arith: int = _lib.ARITH_DISTANCE_RS) -> np.ndarray:
        """Rows of `vectors` normalised on the device (a copy is returned): distance.rs's `normalize` recipe, or
        the accelerator's `batch_normalize` with arith ACCEL_CHUNK4."""
        v = np.array(vectors, dtype=np.float32, order="C", copy=True)
        flat = v.reshape(1, -1) if v.ndim == 1 else v
        if flat.size:
            _lib.check(_lib.lib().needle_hip_normalize(self._h, arith, flat.ctypes.data, flat.shape[0], flat.shape[1]))
        return v

    def normalize_device(self, d_vectors, arith: int = _lib.ARITH_DISTANCE_RS) -> None:
        """In place, on a contiguous float32 [n, dim] device tensor."""
        _wait_for_producer(d_vectors)
        _lib.check(_lib.lib().needle_hip_normalize_device(self._h, arith, d_vectors.data_ptr(), int(d_vectors.shape[0]),
                                                          int(d_vectors.shape[1])))

    def batch_top_k(self, scores, k: int):
        """(idx[rows,k] uint64, scores[rows,k] f32, counts[rows] uint32): the k smallest of every row by (score, index)."""
        s = _f32(scores)
        if s.ndim == 1:
            s = s.reshape(1, -1)
        rows, n = s.shape
        kk = max(k, 1)
        idx = np.full((rows, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        val = np.full((rows, kk), np.inf, dtype=np.float32)
        cnt = np.zeros(rows, dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_batch_top_k(self._h, s.ctypes.data if s.size else None, rows, n, k, idx.ctypes.data,
                                                     val.ctypes.data, cnt.ctypes.data))
        return idx[:, :k], val[:, :k], cnt

    def merge_topk(self, dists, ids, counts, k: int):
        """Host form of the shard merge: dists/ids [lists, q, per], counts [lists, q] -> (ids[q,k], dists[q,k], counts[q])."""
        d, i = _f32(dists), np.ascontiguousarray(ids, dtype=np.uint64)
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        lists, q, per = d.shape
        kk = max(k, 1)
        oi = np.full((q, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        od = np.full((q, kk), np.inf, dtype=np.float32)
        oc = np.zeros(q, dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_merge_topk(self._h, d.ctypes.data, i.ctypes.data, c.ctypes.data, lists, q, per, k,
                                                    oi.ctypes.data, od.ctypes.data, oc.ctypes.data))
        return oi[:, :k], od[:, :k], oc

    def distance_matrix(self, metric: DistanceFunction, queries, base, arith: int = _lib.ARITH_SERVICE) -> np.ndarray:
        """[queries][vectors] distances (the dot product negated), every value in the chosen arithmetic:
        distance_matrix_cosine / _euclidean of src/services/compute/gpu_kernels.rs:223-265 with ARITH_SERVICE."""
        q, b = _f32(queries), _f32(base)
        if q.ndim != 2 or b.ndim != 2 or (q.shape[0] and b.shape[0] and q.shape[1] != b.shape[1]):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "Dimension mismatch")
        out = np.empty((q.shape[0], b.shape[0]), dtype=np.float32)
        if out.size:
            _lib.check(_lib.lib().needle_hip_distance_matrix(self._h, int(metric), arith, q.ctypes.data, q.shape[0],
                                                             b.ctypes.data, b.shape[0], q.shape[1], out.ctypes.data))
        return out

    def distance_matrix_device(self, metric: DistanceFunction, d_queries, d_base, d_out, arith: int = _lib.ARITH_SERVICE) -> None:
        for t in (d_queries, d_base):
            _wait_for_producer(t)
        _lib.check(_lib.lib().needle_hip_distance_matrix_device(self._h, int(metric), arith, d_queries.data_ptr(),
                                                                int(d_queries.shape[0]), d_base.data_ptr(),
                                                                int(d_base.shape[0]), int(d_queries.shape[1]), d_out.data_ptr()))

    def multi_query_search(self, metric: DistanceFunction, queries, base, k: int, arith: int = _lib.ARITH_SERVICE):
        """(idx[q,k] uint64, dists[q,k] f32, counts[q] uint32): per query the k nearest vectors by (distance, index),
        formed on the device from the distance matrix (multi_query_search, gpu_kernels.rs:267-300)."""
        q, b = _f32(queries), _f32(base)
        if q.ndim != 2 or b.ndim != 2 or (q.shape[0] and b.shape[0] and q.shape[1] != b.shape[1]):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "Dimension mismatch")
        nq, kk = q.shape[0], max(k, 1)
        idx = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        val = np.full((nq, kk), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        if nq:
            _lib.check(_lib.lib().needle_hip_multi_query_search(self._h, int(metric), arith, q.ctypes.data, nq,
                                                                b.ctypes.data if b.size else None, b.shape[0], q.shape[1], k,
                                                                idx.ctypes.data, val.ctypes.data, cnt.ctypes.data))
        return idx[:, :k], val[:, :k], cnt

    def merge_topk_device(self, d_dists, d_ids, d_counts, lists: int, q: int, per: int, k: int, d_out_ids, d_out_dists,
                          d_out_counts) -> None:
        _lib.check(_lib.lib().needle_hip_merge_topk_device(self._h, d_dists.data_ptr(), d_ids.data_ptr(),
                                                           d_counts.data_ptr(), lists, q, per, k, d_out_ids.data_ptr(),
                                                           d_out_dists.data_ptr(), d_out_counts.data_ptr()))


class FlatIndex:
    """Resident exact flat index (needle_hip_index, kind flat)."""

    def __init__(self, ctx: Context, base, metric: DistanceFunction, *, arith: int = _lib.ARITH_DISTANCE_RS,
                 valid=None, dim: int | None = None):
        self.ctx = ctx
        self.metric = DistanceFunction(metric)
        self._h = C.c_void_p()
        L = _lib.lib()
        if hasattr(base, "data_ptr"):  # device tensor
            if base.dim() != 2 or str(base.dtype) != "torch.float32" or not base.is_contiguous():
                raise ValueError("device base must be a contiguous float32 [n, dim] tensor")
            n, d = int(base.shape[0]), int(base.shape[1])
            vm = _mask_words(valid, n)
            _wait_for_producer(base)
            _lib.check(L.needle_hip_flat_build_device(ctx._h, base.data_ptr(), n, d, int(self.metric), arith,
                                                      vm.ctypes.data if vm is not None else None, C.byref(self._h)))
        else:
            b = _f32(base)
            if b.ndim != 2:
                if dim is None:
                    raise ValueError("an empty base needs an explicit dim")
                b = b.reshape(0, dim)
            n, d = b.shape
            if n == 0 and dim is not None:
                d = dim
            vm = _mask_words(valid, n)
            _lib.check(L.needle_hip_flat_build(ctx._h, b.ctypes.data if n else None, n, d, int(self.metric), arith,
                                               vm.ctypes.data if vm is not None else None, C.byref(self._h)))
        self.n, self.dim = n, d

    @classmethod
    def from_rows(cls, ctx: Context, rows, metric: DistanceFunction, *, arith: int = _lib.ARITH_DISTANCE_RS, valid=None):
        """Build from separately allocated rows (the reference's Vec<Vec<f32>>)."""
        keep = [_f32(r) for r in rows]
        n = len(keep)
        d = keep[0].size if n else 0
        ptrs = (C.c_void_p * max(n, 1))(*[r.ctypes.data for r in keep])
        self = cls.__new__(cls)
        self.ctx, self.metric, self._h = ctx, DistanceFunction(metric), C.c_void_p()
        vm = _mask_words(valid, n)
        _lib.check(_lib.lib().needle_hip_flat_build_rows(ctx._h, ptrs, n, d, int(metric), arith,
                                                         vm.ctypes.data if vm is not None else None, C.byref(self._h)))
        self.n, self.dim = n, d
        return self

    def close(self) -> None:
        _destroy_handle(self)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(_lib.lib().needle_hip_index_len(self._h))

    def set_valid_mask(self, valid) -> None:
        vm = _mask_words(valid, self.n)
        _lib.check(_lib.lib().needle_hip_index_set_valid_mask(self._h, vm.ctypes.data if vm is not None else None))

    def append(self, rows, valid=None) -> None:
        """Rows appended behind the resident ones (needle_hip_index_append): row j of the batch becomes internal id
        len + j; only the new rows are uploaded.  `rows` is a host array or a contiguous float32 device tensor."""
        if hasattr(rows, "data_ptr"):
            if rows.dim() != 2 or str(rows.dtype) != "torch.float32" or not rows.is_contiguous() or int(rows.shape[1]) != self.dim:
                raise ValueError("device rows must be a contiguous float32 [n, dim] tensor")
            m = int(rows.shape[0])
            vm = _mask_words(valid, m)
            _wait_for_producer(rows)
            _lib.check(_lib.lib().needle_hip_index_append_device(self._h, rows.data_ptr(), m,
                                                                  vm.ctypes.data if vm is not None else None))
        else:
            r = _f32(rows)
            if r.ndim == 1:
                r = r.reshape(1, -1)
            if r.ndim != 2 or r.shape[1] != self.dim:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                             f"Dimension mismatch: expected {self.dim}, got {r.shape[-1]}")
            m = int(r.shape[0])
            vm = _mask_words(valid, m)
            _lib.check(_lib.lib().needle_hip_index_append(self._h, r.ctypes.data, m, vm.ctypes.data if vm is not None else None))
        self.n += m

    def swap_remove(self, pos: int) -> None:
        """The last row takes internal id `pos`; the index shrinks by one (Vec::swap_remove, needle_hip_index_swap_remove)."""
        _lib.check(_lib.lib().needle_hip_index_swap_remove(self._h, int(pos)))
        self.n -= 1

    def search(self, queries, k: int, *, precision: str = "fp32", filter_mask=None):
        """Host arrays in, host arrays out: (ids[q,k] uint64, dists[q,k] float32, counts[q] uint32)."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.ndim != 2:
            raise ValueError("queries must be [q, dim]")
        nq, qd = q.shape
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fm = _mask_words(filter_mask, self.n)
        sp = _lib.SearchParams(k=k, nprobe=0, precision=PRECISION[precision], reserved=0,
                               filter_bitmask=fm.ctypes.data if fm is not None else None)
        _lib.check(_lib.lib().needle_hip_batch_search(self._h, q.ctypes.data if nq else None, nq, qd, C.byref(sp),
                                                      ids.ctypes.data, dists.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dists[:, :k], counts

    def search_device(self, d_queries, k: int, d_out_ids, d_out_dists, d_out_counts, *, precision: str = "fp32",
                      d_filter=None) -> None:
        """Device tensors in and out (float32 [q,dim]; int64/uint64 [q,k]; float32 [q,k]; int32/uint32 [q])."""
        nq = int(d_queries.shape[0])
        _wait_for_producer(d_queries)
        if int(d_queries.shape[1]) != self.dim:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dim}, got {int(d_queries.shape[1])}")
        sp = _lib.SearchParams(k=k, nprobe=0, precision=PRECISION[precision], reserved=0,
                               filter_bitmask=d_filter.data_ptr() if d_filter is not None else None)
        _lib.check(_lib.lib().needle_hip_batch_search_device(self._h, d_queries.data_ptr(), nq, C.byref(sp),
                                                             d_out_ids.data_ptr(), d_out_dists.data_ptr(),
                                                             d_out_counts.data_ptr()))
