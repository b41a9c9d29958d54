"""Host-side mirror of the reference's `Collection` search surface
(src/collection/{batch,search_methods,validation}.rs) with the exact flat GPU
scan behind it: `search`, `batch_search`, `batch_search_with_filter`, keyed by
the same `DistanceFunction`, returning `SearchResult{id, distance, metadata}`
ascending by (distance, internal id).

What differs from the reference on purpose: the reference's `batch_search`
walks an HNSW graph (approximate); this one is the exact scan that its
`brute_force_search` / `compute_ground_truth` define, keeping the k NEAREST
rows (the reference's brute-force heap evicts the nearest).
Filters are applied as an exact pre-filter, not as a post-filter of an
over-fetched candidate list.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .distance import DistanceFunction
from .index import Context, FlatIndex, _f32


@dataclasses.dataclass
class SearchResult:  # src/collection/search.rs:39-52
    id: str
    distance: float
    metadata: Optional[Any] = None


class Collection:
    def __init__(self, name: str, dimensions: int, distance: DistanceFunction = DistanceFunction.Cosine, *,
                 device: int = 0, ctx: Context | None = None, precision: str = "fp32"):
        self.name = name
        self.dimensions = dimensions
        self.distance = DistanceFunction(distance)
        self.precision = precision
        self.ctx = ctx or Context(device)
        self._vectors: List[np.ndarray] = []     # internal id -> row (VectorStore)
        self._external: List[str] = []           # internal id -> external id
        self._metadata: List[Optional[Any]] = []
        self._by_external: Dict[str, int] = {}
        self._deleted: List[bool] = []
        self._index: FlatIndex | None = None
        self._dirty = True
        self._mask_dirty = False

    # ---- validation.rs -------------------------------------------------------
    @staticmethod
    def validate_vector(vector: Sequence[float]) -> None:  # validation.rs:8-24
        for i, v in enumerate(vector):
            if math.isnan(v):
                raise _lib.InvalidVector(_lib.ERR_INVALID_VECTOR, f"Invalid vector: Vector contains NaN at index {i}")
            if math.isinf(v):
                raise _lib.InvalidVector(_lib.ERR_INVALID_VECTOR, f"Invalid vector: Vector contains Inf at index {i}")

    def validate_query(self, query: Sequence[float]) -> None:  # validation.rs:27-35
        if len(query) != self.dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dimensions}, got {len(query)}")
        self.validate_vector(query)

    def clamp_k(self, k: int) -> int:  # validation.rs:39-41
        return min(k, len(self))

    def __len__(self) -> int:
        return len(self._vectors) - sum(self._deleted)

    # ---- mutation ------------------------------------------------------------
    def insert(self, id: str, vector: Sequence[float], metadata: Optional[Any] = None) -> None:
        self.validate_query(vector)  # validate_insert_input, validation.rs:44-52
        if id in self._by_external:
            raise ValueError(f"Vector already exists: {id}")
        self._by_external[id] = len(self._vectors)
        self._vectors.append(_f32(vector))
        self._external.append(id)
        self._metadata.append(metadata)
        self._deleted.append(False)
        self._dirty = True

    def delete(self, id: str) -> bool:
        i = self._by_external.pop(id, None)
        if i is None:
            return False
        self._deleted[i] = True
        self._mask_dirty = True
        return True

    def _resident(self) -> FlatIndex:
        if self._dirty or self._index is None:
            if self._index is not None:
                self._index.close()
            base = np.stack(self._vectors) if self._vectors else np.zeros((0, self.dimensions), dtype=np.float32)
            self._index = FlatIndex(self.ctx, base, self.distance, dim=self.dimensions,
                                    valid=None if not any(self._deleted) else ~np.array(self._deleted))
            self._dirty = self._mask_dirty = False
        elif self._mask_dirty:
            self._index.set_valid_mask(~np.array(self._deleted) if any(self._deleted) else None)
            self._mask_dirty = False
        return self._index

    # ---- search --------------------------------------------------------------
    def _enrich(self, ids, dists, count) -> List[SearchResult]:  # enrich_results, search_methods.rs:850-866
        return [SearchResult(self._external[int(ids[j])], float(dists[j]), self._metadata[int(ids[j])])
                for j in range(int(count))]

    def search(self, query: Sequence[float], k: int) -> List[SearchResult]:
        return self.batch_search([query], k)[0]

    def batch_search(self, queries: Sequence[Sequence[float]], k: int) -> List[List[SearchResult]]:
        """Collection::batch_search, src/collection/batch.rs:32-75."""
        for q in queries:  # the whole batch is validated before any search runs
            self.validate_query(q)
        k = self.clamp_k(k)
        if k == 0:
            return [[] for _ in queries]
        if len(queries) == 0:
            return []
        ids, dists, counts = self._resident().search(_f32(queries), k, precision=self.precision)
        return [self._enrich(ids[i], dists[i], counts[i]) for i in range(len(queries))]

    def batch_search_with_filter(self, queries: Sequence[Sequence[float]], k: int,
                                 filter: Callable[[Optional[Any]], bool]) -> List[List[SearchResult]]:
        """Collection::batch_search_with_filter, src/collection/batch.rs:108-171; `filter` plays
        Filter::matches(metadata).  Rows without a matching metadata entry are excluded
        (search_methods.rs:215-223)."""
        for q in queries:
            self.validate_query(q)
        k = self.clamp_k(k)
        if k == 0:
            return [[] for _ in queries]
        if len(queries) == 0:
            return []
        keep = np.array([bool(filter(m)) for m in self._metadata], dtype=np.bool_)
        ids, dists, counts = self._resident().search(_f32(queries), k, precision=self.precision, filter_mask=keep)
        return [self._enrich(ids[i], dists[i], counts[i]) for i in range(len(queries))]
