"""Host-side mirror of the reference's operator API, `GpuAccelerator`
(src/experimental/gpu/mod.rs:308-1080): same method names, argument meaning,
result shapes and error behaviour, served by libneedle_hip.so.

Arithmetic: these operators reproduce the accelerator's own scalar order
(groups of four, src/experimental/gpu/common.rs:515-581), not distance.rs.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import _lib
from .distance import DistanceFunction, DistanceType
from .index import Context, FlatIndex, _f32

Hit = Tuple[int, float]


class GpuAccelerator:
    def __init__(self, device: int = 0, ctx: Context | None = None):
        self.ctx = ctx or Context(device)

    # -- batch_*_distance, gpu/mod.rs:308-448 ---------------------------------
    def batch_cosine_distance(self, query, vectors) -> np.ndarray:
        return self._batch_distance(DistanceFunction.Cosine, query, vectors)

    def batch_euclidean_distance(self, query, vectors) -> np.ndarray:
        return self._batch_distance(DistanceFunction.Euclidean, query, vectors)

    def batch_dot_product(self, query, vectors) -> np.ndarray:
        return self._batch_distance(DistanceFunction.DotProduct, query, vectors)

    def _batch_distance(self, metric, query, vectors) -> np.ndarray:
        if len(vectors) == 0:  # gpu/mod.rs:315-317
            return np.empty(0, dtype=np.float32)
        q = _f32(query)
        if any(len(v) != q.size for v in vectors):  # gpu/mod.rs:319-322 -> Err("Dimension mismatch")
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "Dimension mismatch")
        return self.ctx.batch_distance(metric, q, _f32(vectors))

    # -- fused_search / batch_fused_search, gpu/mod.rs:706-823 ----------------
    def fused_search(self, query, vectors, k: int, distance_type: DistanceType) -> List[Hit]:
        res = self.batch_fused_search([query], vectors, k, distance_type)
        return res[0] if res else []

    def batch_fused_search(self, queries, vectors, k: int, distance_type: DistanceType) -> List[List[Hit]]:
        if len(queries) == 0 or len(vectors) == 0:  # gpu/mod.rs:799-801
            return []
        q, b = _f32(queries), _f32(vectors)
        ix = FlatIndex(self.ctx, b, DistanceType(distance_type).metric(), arith=_lib.ARITH_ACCEL_CHUNK4)
        try:
            ids, dists, counts = ix.search(q, min(k, b.shape[0]))
        finally:
            ix.close()
        return [[(int(ids[i, j]), float(dists[i, j])) for j in range(int(counts[i]))] for i in range(q.shape[0])]

    # -- streaming_batch_search, gpu/mod.rs:855-903 ---------------------------
    def streaming_batch_search(self, query, vectors, k: int, distance_type: DistanceType,
                               chunk_size: int | None = None) -> List[Hit]:
        """Chunked search with a global merge.  The device holds the whole base, so the chunking of the
        reference (a host-memory measure) collapses into one resident search with the same result set."""
        if len(vectors) == 0:
            return []
        return self.fused_search(query, vectors, k, distance_type)

    # -- prefiltered_batch_search, gpu/mod.rs:1033-1080 -----------------------
    def prefiltered_batch_search(self, query, vectors: Sequence[Tuple[int, Sequence[float]]], k: int,
                                 distance_type: DistanceType, filter: Callable[[int], bool]) -> List[Hit]:
        keep = np.array([bool(filter(idx)) for idx, _ in vectors], dtype=np.bool_)
        if not keep.any():
            return []
        b = _f32([v for _, v in vectors])
        original = [idx for idx, _ in vectors]
        ix = FlatIndex(self.ctx, b, DistanceType(distance_type).metric(), arith=_lib.ARITH_ACCEL_CHUNK4)
        try:
            ids, dists, counts = ix.search(_f32(query).reshape(1, -1), min(k, int(keep.sum())), filter_mask=keep)
        finally:
            ix.close()
        return [(original[int(ids[0, j])], float(dists[0, j])) for j in range(int(counts[0]))]

    # -- kmeans_assign, gpu/mod.rs:660-703 ------------------------------------
    def kmeans_assign(self, vectors, centroids) -> List[int]:
        return self.ctx.kmeans_assign(vectors, centroids, _lib.ARITH_ACCEL_CHUNK4).tolist()

    # -- batch_top_k, gpu/mod.rs:534-549 --------------------------------------
    def metrics(self) -> dict:
        return self.ctx.metrics()


class GpuResidentIndex:
    """`GpuResidentIndex` (src/experimental/gpu/mod.rs:1102-1213): vectors stay on the device between
    searches; `add_batch` marks the index dirty and the next search re-uploads."""

    def __init__(self, dimensions: int, distance_type: DistanceType = DistanceType.Cosine, device: int = 0,
                 ctx: Context | None = None):
        self.ctx = ctx or Context(device)
        self.dimensions = dimensions
        self.distance_type = DistanceType(distance_type)
        self._rows: list[np.ndarray] = []
        self._ids: list[int] = []
        self._index: FlatIndex | None = None
        self.dirty = False

    def __len__(self) -> int:
        return len(self._rows)

    def add(self, id: int, vector) -> None:
        v = _f32(vector)
        if v.size != self.dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dimensions}, got {v.size}")
        self._rows.append(v)
        self._ids.append(id)
        self.dirty = True

    def add_batch(self, items) -> None:
        for id, v in items:
            self.add(id, v)

    def remove(self, id: int) -> bool:
        try:
            i = self._ids.index(id)
        except ValueError:
            return False
        del self._ids[i]
        del self._rows[i]
        self.dirty = True
        return True

    def search(self, query, k: int) -> List[Hit]:
        if not self._rows:
            return []
        if self.dirty or self._index is None:
            if self._index is not None:
                self._index.close()
            self._index = FlatIndex(self.ctx, np.stack(self._rows), self.distance_type.metric(),
                                    arith=_lib.ARITH_ACCEL_CHUNK4)
            self.dirty = False
        ids, dists, counts = self._index.search(_f32(query).reshape(1, -1), min(k, len(self._rows)))
        return [(self._ids[int(ids[0, j])], float(dists[0, j])) for j in range(int(counts[0]))]
