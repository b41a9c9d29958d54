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
    MERGE_ENTRIES = 3750  # (distance, id) entries of one query that needle_hip_merge_topk_device merges per call

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
        """Chunked search with a global merge: every chunk of `chunk_size` rows (default: the reference's
        calculate_optimal_chunk_size, gpu/mod.rs:954-973) is searched for its own top k, the per-chunk lists are
        concatenated with chunk-relative indices made global and merged by (distance, index) on the device
        (gpu/mod.rs:876-892: concat, stable sort by distance, truncate)."""
        n = len(vectors)
        if n == 0:
            return []
        b = _f32(vectors)
        if chunk_size is None:
            # half of the device memory (288 GB) in rows, clamped to [100, 50000] rows
            chunk_size = int(min(max((144 << 30) // max(b.shape[1] * 4, 1), 100), 50_000))
        chunk_size = max(int(chunk_size), 1)
        if n <= chunk_size:
            return self.fused_search(query, b, k, distance_type)
        q = _f32(query).reshape(1, -1)
        starts = list(range(0, n, chunk_size))
        kk = max(k, 1)
        # the device merge takes MERGE_ENTRIES (distance, id) entries of one query per call: the chunks' lists are folded
        # into a running top k, a group at a time (top-k merging is associative, so the grouping does not change the
        # result); a k too large for even two lists per call is served by one resident search, which returns the same
        group = self.MERGE_ENTRIES // kk - 1
        if group < 1:
            return self.fused_search(query, b, k, distance_type)
        metric = DistanceType(distance_type).metric()
        umax = np.iinfo(np.uint64).max
        run_i = np.full((1, 1, kk), umax, dtype=np.uint64)
        run_d = np.full((1, 1, kk), np.inf, dtype=np.float32)
        run_c = np.zeros((1, 1), dtype=np.uint32)
        for g0 in range(0, len(starts), group):
            part = starts[g0:g0 + group]
            ids = np.full((len(part), 1, kk), umax, dtype=np.uint64)
            dists = np.full((len(part), 1, kk), np.inf, dtype=np.float32)
            counts = np.zeros((len(part), 1), dtype=np.uint32)
            for c, s in enumerate(part):
                ix = FlatIndex(self.ctx, b[s:s + chunk_size], metric, arith=_lib.ARITH_ACCEL_CHUNK4)
                try:
                    ci, cd, cc = ix.search(q, min(k, min(chunk_size, n - s)))
                finally:
                    ix.close()
                m = int(cc[0])
                ids[c, 0, :m] = ci[0, :m] + np.uint64(s)
                dists[c, 0, :m] = cd[0, :m]
                counts[c, 0] = m
            oi, od, oc = self.ctx.merge_topk(np.concatenate([run_d, dists]), np.concatenate([run_i, ids]),
                                             np.concatenate([run_c, counts]), k)
            m = int(oc[0])
            run_i[0, 0, :m], run_d[0, 0, :m], run_c[0, 0] = oi[0, :m], od[0, :m], m
        return [(int(run_i[0, 0, j]), float(run_d[0, 0, j])) for j in range(int(run_c[0, 0]))]

    # -- streaming_multi_query_search, gpu/mod.rs:906-951 ----------------------
    def streaming_multi_query_search(self, queries, vectors, k: int, distance_type: DistanceType) -> List[List[Hit]]:
        """Every query's streaming_batch_search result.  The chunked search's merge (concatenate the chunks' lists in
        chunk order, stable sort by distance, truncate) yields the k nearest by (distance, index) over all vectors, so
        the whole batch is one pass here: distance rows formed on the device a block of queries at a time, each block
        through the batch_top_k selection."""
        nq = len(queries)
        if nq == 0 or len(vectors) == 0:  # gpu/mod.rs:915-917
            return [[] for _ in range(nq)]
        q, b = _f32(queries), _f32(vectors)
        idx, val, cnt = self.ctx.multi_query_search(DistanceType(distance_type).metric(), q, b, min(k, b.shape[0]),
                                                    _lib.ARITH_ACCEL_CHUNK4)
        return [[(int(idx[i, j]), float(val[i, j])) for j in range(int(cnt[i]))] for i in range(nq)]

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
    def batch_top_k(self, distances, k: int) -> List[Hit]:
        """The k smallest of `distances` as (index, distance), ascending; equal distances keep index order."""
        if len(distances) == 0 or k == 0:
            return []
        idx, val, cnt = self.ctx.batch_top_k(_f32(distances).reshape(1, -1), k)
        return [(int(idx[0, j]), float(val[0, j])) for j in range(int(cnt[0]))]

    # -- batch_normalize, gpu/mod.rs:451-473 ----------------------------------
    def batch_normalize(self, vectors) -> np.ndarray:
        """Every vector divided by its norm (normalize_vector_simd, gpu/common.rs:628-635); returns the result."""
        if len(vectors) == 0:
            return np.empty((0, 0), dtype=np.float32)
        return self.ctx.normalize(_f32(vectors), _lib.ARITH_ACCEL_CHUNK4)

    def metrics(self) -> dict:
        return self.ctx.metrics()


class FallbackSearchResult:
    """`FallbackSearchResult`, src/experimental/gpu/mod.rs:1398-1403."""

    __slots__ = ("results", "backend_used", "elapsed")

    def __init__(self, results: List[Hit], backend_used: str, elapsed: float):
        self.results, self.backend_used, self.elapsed = results, backend_used, elapsed


class TransparentFallbackManager:
    """The reference's failure bookkeeping around one-query searches (`TransparentFallbackManager`,
    src/experimental/gpu/mod.rs:1407-1560): every device error counts, from the fourth on the device path is switched off
    for `backoff_secs` (5 s, doubling up to 300 s), and `gpu_reliability` is successes over attempts.

    The reference answers a failed or switched-off call from its own scalar loop.  This package holds no host
    implementation of any search: the caller's existing path is passed in as `cpu_search(query, vectors, k, distance)`
    (Needle's own `cpu_search`, mod.rs:1496-1541), and without one the device error is raised to the caller."""

    GPU, CPU = "Gpu", "CpuSimd"  # ExecutionBackend variants the reference reports

    def __init__(self, accelerator: GpuAccelerator, cpu_search: Callable | None = None, clock: Callable[[], float] | None = None):
        import time

        self.accelerator = accelerator
        self.cpu_search = cpu_search
        self._clock = clock or time.monotonic
        self.reset()

    def reset(self) -> None:  # mod.rs:1554-1559
        self.gpu_failures = 0
        self.gpu_successes = 0
        self.gpu_disabled_until: float | None = None
        self.backoff_secs = 5

    def gpu_reliability(self) -> float:  # mod.rs:1544-1551
        total = self.gpu_successes + self.gpu_failures
        return 1.0 if total == 0 else self.gpu_successes / total

    def _try_gpu_search(self, query, vectors, k: int, distance: DistanceType) -> List[Hit]:  # mod.rs:1476-1494
        fn = {DistanceType.Cosine: self.accelerator.batch_cosine_distance,
              DistanceType.Euclidean: self.accelerator.batch_euclidean_distance,
              DistanceType.DotProduct: self.accelerator.batch_dot_product}[DistanceType(distance)]
        d = fn(query, vectors)
        # sort_by(partial_cmp ... unwrap_or(Equal)) is a stable sort: equal distances keep their index order
        order = sorted(range(len(d)), key=lambda i: d[i]) if not np.isnan(d).any() else _partial_cmp_sort(d)
        return [(int(i), float(d[i])) for i in order[:k]]

    def search(self, query, vectors, k: int, distance: DistanceType) -> FallbackSearchResult:  # mod.rs:1427-1474
        start = self._clock()
        gpu_available = True
        if self.gpu_disabled_until is not None:
            if self._clock() < self.gpu_disabled_until:
                gpu_available = False
            else:
                self.gpu_disabled_until = None
        error = None
        if gpu_available:
            try:
                results = self._try_gpu_search(query, vectors, k, distance)
                self.gpu_successes += 1
                return FallbackSearchResult(results, self.GPU, self._clock() - start)
            except _lib.NeedleHipError as e:
                error = e
                self.gpu_failures += 1
                if self.gpu_failures > 3:
                    self.gpu_disabled_until = self._clock() + self.backoff_secs
                    self.backoff_secs = min(self.backoff_secs * 2, 300)
        if self.cpu_search is None:
            raise error if error is not None else _lib.NeedleHipError(
                _lib.ERR_NO_DEVICE, "device path switched off after repeated failures and no cpu_search was given")
        results = self.cpu_search(query, vectors, k, distance)
        return FallbackSearchResult(list(results), self.CPU, self._clock() - start)


def _partial_cmp_sort(d) -> List[int]:
    """Rust's `sort_by(|a, b| a.partial_cmp(b).unwrap_or(Equal))` with NaNs present: a stable merge sort under a
    comparison that calls NaN equal to everything (insertion by that comparison reproduces small inputs; the order of a
    non-transitive comparison is otherwise unspecified in the reference too)."""
    import functools

    def cmp(i, j):
        a, b = d[i], d[j]
        return -1 if a < b else (1 if a > b else 0)

    return sorted(range(len(d)), key=functools.cmp_to_key(cmp))


class GpuResidentIndex:
    """`GpuResidentIndex` (src/experimental/gpu/mod.rs:1102-1213): vectors stay on the device between searches.
    `add` / `add_batch` queue rows on the host; the next search appends the queued rows to the resident index
    (needle_hip_index_append: only the new rows cross PCIe).  `remove` is the reference's `swap_remove`: the last
    vector takes the removed one's position (needle_hip_index_swap_remove, a row copy on the device)."""

    def __init__(self, dimensions: int, distance_type: DistanceType = DistanceType.Cosine, device: int = 0,
                 ctx: Context | None = None):
        self.ctx = ctx or Context(device)
        self.dimensions = dimensions
        self.distance_type = DistanceType(distance_type)
        self._ids: list[int] = []             # external id by position (resident rows first, then the queued ones)
        self._pending: list[np.ndarray] = []  # rows added since the last search
        self._index: FlatIndex | None = None
        self.dirty = False

    def __len__(self) -> int:
        return len(self._ids)

    def add(self, id: int, vector) -> None:
        v = _f32(vector)
        if v.size != self.dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dimensions}, got {v.size}")
        self._pending.append(v)
        self._ids.append(id)
        self.dirty = True

    def add_batch(self, items) -> None:
        for id, v in items:
            self.add(id, v)

    def remove(self, id: int) -> bool:
        try:
            pos = self._ids.index(id)
        except ValueError:
            return False
        self._sync()  # positions refer to the resident order
        self._ids[pos] = self._ids[-1]
        self._ids.pop()
        self._index.swap_remove(pos)
        return True

    def _sync(self) -> None:
        """Queued rows go to the device: an append, or the first build."""
        if self._pending:
            rows = np.stack(self._pending)
            self._pending = []
            if self._index is None:
                self._index = FlatIndex(self.ctx, rows, self.distance_type.metric(), arith=_lib.ARITH_ACCEL_CHUNK4)
            else:
                self._index.append(rows)
        self.dirty = False

    def search(self, query, k: int) -> List[Hit]:
        if not self._ids:
            return []
        self._sync()
        ids, dists, counts = self._index.search(_f32(query).reshape(1, -1), min(k, len(self._ids)))
        return [(self._ids[int(ids[0, j])], float(dists[0, j])) for j in range(int(counts[0]))]
