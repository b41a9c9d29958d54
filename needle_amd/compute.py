"""Host-side mirror of the compute service's accelerator (src/services/compute/gpu_kernels.rs:77-359): batch
distances, the queries x vectors distance matrix and multi-query search, served by libneedle_hip.so in that file's
own arithmetic (its cosine is one loop over three sums from +0.0 with `denominator < f32::EPSILON -> 1.0`,
:345-350; its euclidean and dot product are plain f32 sums, :352-358).  In the reference every method of this
type runs on the CPU whatever the backend label says; here they run on the device and return the same values.
"""
from __future__ import annotations

import dataclasses
import time
from typing import List, Tuple

import numpy as np

from . import _lib
from .distance import DistanceFunction
from .index import Context, _f32

BACKEND = "HIP"  # GpuBackend has no AMD variant (gpu_kernels.rs:28-36); results carry this label instead


@dataclasses.dataclass
class BatchDistanceResult:  # gpu_kernels.rs:64-75
    distances: np.ndarray
    backend: str
    compute_us: int
    throughput: float


@dataclasses.dataclass
class DistanceMatrixResult:  # gpu_kernels.rs:96-130
    matrix: np.ndarray  # row-major [num_queries * num_vectors]
    num_queries: int
    num_vectors: int
    backend: str
    compute_us: int

    def get(self, q: int, v: int) -> float:
        return float(self.matrix[q * self.num_vectors + v])

    def top_k(self, query_idx: int, k: int) -> List[Tuple[int, float]]:
        """Stable sort of the query's row, truncated to k (:118-129): equal distances keep index order."""
        row = self.matrix[query_idx * self.num_vectors:(query_idx + 1) * self.num_vectors]
        order = np.argsort(row, kind="stable")[:k]
        return [(int(i), float(row[i])) for i in order]


@dataclasses.dataclass
class MultiQueryResult:  # gpu_kernels.rs:133-143
    results: List[List[Tuple[int, float]]]
    backend: str
    compute_us: int
    queries_per_second: float


def _rate(count: int, us: int) -> float:
    return count / (us / 1e6) if us > 0 else 0.0


class GpuAccelerator:
    def __init__(self, device: int = 0, ctx: Context | None = None, min_batch_size: int = 100):
        self.ctx = ctx or Context(device)
        self.min_batch_size = min_batch_size
        self.total_computations = 0

    def backend(self) -> str:
        return BACKEND

    def is_gpu_available(self) -> bool:
        return True

    def _matrix(self, metric, queries, vectors) -> Tuple[np.ndarray, int]:
        t0 = time.perf_counter()
        q, b = _f32(queries), _f32(vectors)
        if b.size == 0 or q.size == 0:
            out = np.empty((q.shape[0] if q.ndim == 2 else 0, b.shape[0] if b.ndim == 2 else 0), dtype=np.float32)
        else:
            out = self.ctx.distance_matrix(metric, q, b, _lib.ARITH_SERVICE)
        return out, int((time.perf_counter() - t0) * 1e6)

    def _batch(self, metric, query, vectors) -> BatchDistanceResult:
        m, us = self._matrix(metric, _f32(query).reshape(1, -1), vectors)
        d = m.reshape(-1)
        return BatchDistanceResult(d, BACKEND, us, _rate(d.size, us))

    def batch_cosine_distance(self, query, vectors) -> BatchDistanceResult:  # :172-181
        return self._batch(DistanceFunction.Cosine, query, vectors)

    def batch_euclidean_distance(self, query, vectors) -> BatchDistanceResult:  # :184-193
        return self._batch(DistanceFunction.Euclidean, query, vectors)

    def batch_dot_product(self, query, vectors) -> BatchDistanceResult:  # :196-205, the NEGATED dot product
        return self._batch(DistanceFunction.DotProduct, query, vectors)

    def hnsw_neighbor_distances(self, node_vector, candidates) -> BatchDistanceResult:  # :304-326
        return self._batch(DistanceFunction.Cosine, node_vector, candidates)

    def distance_matrix_cosine(self, queries, vectors) -> DistanceMatrixResult:  # :223-243
        m, us = self._matrix(DistanceFunction.Cosine, queries, vectors)
        return DistanceMatrixResult(m.reshape(-1), m.shape[0], m.shape[1], BACKEND, us)

    def distance_matrix_euclidean(self, queries, vectors) -> DistanceMatrixResult:  # :246-265
        m, us = self._matrix(DistanceFunction.Euclidean, queries, vectors)
        return DistanceMatrixResult(m.reshape(-1), m.shape[0], m.shape[1], BACKEND, us)

    def multi_query_search(self, queries, vectors, k: int) -> MultiQueryResult:  # :267-300 (cosine)
        t0 = time.perf_counter()
        q, b = _f32(queries), _f32(vectors)
        nq = q.shape[0] if q.ndim == 2 else 0
        if nq == 0:
            results: List[List[Tuple[int, float]]] = []
        elif b.size == 0:
            results = [[] for _ in range(nq)]
        else:
            idx, val, cnt = self.ctx.multi_query_search(DistanceFunction.Cosine, q, b, min(k, b.shape[0]), _lib.ARITH_SERVICE)
            results = [[(int(idx[i, j]), float(val[i, j])) for j in range(int(cnt[i]))] for i in range(nq)]
        us = int((time.perf_counter() - t0) * 1e6)
        return MultiQueryResult(results, BACKEND, us, _rate(nq, us))
