"""Host-side mirror of the reference's `Collection` search surface
(src/collection/{batch,search_methods,validation}.rs) with the exact flat GPU
scan behind it: `search`, `batch_search`, `batch_search_with_filter`, keyed by
the same `DistanceFunction`, returning `SearchResult{id, distance, metadata}`
ascending by (distance, internal id).

What differs from the reference on purpose: the reference's `batch_search`
walks an HNSW graph (approximate); this one is the exact scan that its
`brute_force_search` / `compute_ground_truth` define, keeping the k NEAREST
rows (the reference's brute-force heap, src/collection/search_methods.rs:189-271,
is a max-heap keyed the wrong way round for n > k: it evicts the nearest row, a
behaviour none of its tests pins; the ground-truth routine of
src/recall_benchmark.rs:183-207 keeps the nearest, and that is what is built here).
Filters are applied as an exact pre-filter, not as a post-filter of an
over-fetched candidate list: `Filter` (needle_amd/metadata.py) is evaluated on
the host into a row bitmask (equality on indexed fields through the inverted
index), deleted and expired rows (src/collection/ttl.rs) are masked the same
way, and the GPU scan never scores a masked row.
"""
from __future__ import annotations

import dataclasses
import math
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .distance import DistanceFunction
from .index import Context, FlatIndex, _f32
from .metadata import Filter, MetadataStore


@dataclasses.dataclass
class SearchResult:  # src/collection/search.rs:39-52
    id: str
    distance: float
    metadata: Optional[Any] = None


def _same_mask(a: Optional[np.ndarray], b: Optional[np.ndarray]) -> bool:
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and bool(np.array_equal(a, b))


class Collection:
    def __init__(self, name: str, dimensions: int, distance: DistanceFunction = DistanceFunction.Cosine, *,
                 device: int = 0, ctx: Context | None = None, precision: str = "fp32",
                 default_ttl_seconds: Optional[int] = None, lazy_expiration: bool = True,
                 clock: Callable[[], float] = time.time):
        self.name = name
        self.dimensions = dimensions
        self.distance = DistanceFunction(distance)
        self.precision = precision
        self._ctx, self._device = ctx, device   # the device context is created on the first search
        self.default_ttl_seconds = default_ttl_seconds   # CollectionConfig::default_ttl_seconds
        self.lazy_expiration = lazy_expiration           # CollectionConfig::lazy_expiration (config.rs:396, default true)
        self._clock = clock                              # Collection::now_unix, ttl.rs:8-13 (injectable for tests)
        self._vectors: List[np.ndarray] = []     # internal id -> row (VectorStore)
        self._external: List[str] = []           # internal id -> external id
        self._metadata: List[Optional[Any]] = []
        self._by_external: Dict[str, int] = {}
        self._deleted: List[bool] = []
        self._store = MetadataStore()            # entries + equality inverted index (metadata.rs)
        self._expirations: Dict[int, int] = {}   # internal id -> unix second it expires at (ttl.rs)
        self._index: FlatIndex | None = None
        self._dirty = True
        self._mask_dirty = False
        self._mask_sent: Optional[np.ndarray] = None

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
    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = Context(self._device)
        return self._ctx

    def now_unix(self) -> int:
        return int(self._clock())

    def insert(self, id: str, vector: Sequence[float], metadata: Optional[Any] = None) -> None:
        self.insert_with_ttl(id, vector, metadata, self.default_ttl_seconds)  # insert_vec, insert.rs:103-110

    def insert_with_ttl(self, id: str, vector: Sequence[float], metadata: Optional[Any] = None,
                        ttl_seconds: Optional[int] = None) -> None:  # insert.rs:89-97, 116-
        self.validate_query(vector)  # validate_insert_input, validation.rs:44-52
        if id in self._by_external:
            raise ValueError(f"Vector already exists: {id}")
        internal = len(self._vectors)
        self._store.insert(internal, id, metadata)
        self._by_external[id] = internal
        self._vectors.append(_f32(vector))
        self._external.append(id)
        self._metadata.append(metadata)
        self._deleted.append(False)
        if ttl_seconds is not None:
            self._expirations[internal] = self.now_unix() + int(ttl_seconds)
        self._dirty = True

    def _drop(self, internal: int) -> None:
        self._by_external.pop(self._external[internal], None)
        self._store.delete(internal)
        self._expirations.pop(internal, None)
        self._deleted[internal] = True
        self._mask_dirty = True

    def delete(self, id: str) -> bool:
        i = self._by_external.get(id)
        if i is None:
            return False
        self._drop(i)
        return True

    # ---- TTL: src/collection/ttl.rs ---------------------------------------------
    def is_expired(self, internal_id: int) -> bool:  # ttl.rs:17-23
        exp = self._expirations.get(internal_id)
        return exp is not None and self.now_unix() >= exp

    def expire_vectors(self) -> int:  # ttl.rs:52-78: the eager sweep
        now = self.now_unix()
        gone = [i for i, exp in self._expirations.items() if now >= exp]
        for i in gone:
            self._drop(i)
        return len(gone)

    def needs_expiration_sweep(self, threshold: float) -> bool:  # ttl.rs:101-115
        if not self._expirations:
            return False
        now = self.now_unix()
        expired = sum(1 for exp in self._expirations.values() if now >= exp)
        total = len(self)
        return expired > 0 if total == 0 else (expired / total) > threshold

    def ttl_stats(self) -> Tuple[int, int, Optional[int], Optional[int]]:  # ttl.rs:130-138
        now = self.now_unix()
        vals = list(self._expirations.values())
        return (len(vals), sum(1 for e in vals if now >= e), min(vals) if vals else None, max(vals) if vals else None)

    def get_ttl(self, id: str) -> Optional[int]:  # ttl.rs:143-146
        i = self._by_external.get(id)
        return None if i is None else self._expirations.get(i)

    def set_ttl(self, id: str, ttl_seconds: Optional[int]) -> None:  # ttl.rs:158-175
        i = self._by_external.get(id)
        if i is None:
            raise KeyError(f"Vector not found: {id}")  # NeedleError::VectorNotFound
        if ttl_seconds is None:
            self._expirations.pop(i, None)
        else:
            self._expirations[i] = self.now_unix() + int(ttl_seconds)

    # ---- resident index and its live-row mask --------------------------------------
    def _live_mask(self) -> Optional[np.ndarray]:
        """Rows the scan may score: not deleted (search_methods.rs:205) and, with lazy expiration, not
        expired at this moment (search_methods.rs:210); None when every row is live."""
        expired = []
        if self.lazy_expiration and self._expirations:
            now = self.now_unix()
            expired = [i for i, exp in self._expirations.items() if now >= exp]
        if not expired and not any(self._deleted):
            return None
        live = ~np.array(self._deleted, dtype=np.bool_)
        live[expired] = False
        return live

    def _resident(self) -> FlatIndex:
        live = self._live_mask()
        if self._dirty or self._index is None:
            if self._index is not None:
                self._index.close()
            base = np.stack(self._vectors) if self._vectors else np.zeros((0, self.dimensions), dtype=np.float32)
            self._index = FlatIndex(self.ctx, base, self.distance, dim=self.dimensions, valid=live)
            self._dirty = self._mask_dirty = False
            self._mask_sent = live
        elif self._mask_dirty or not _same_mask(live, self._mask_sent):
            self._index.set_valid_mask(live)
            self._mask_dirty = False
            self._mask_sent = live
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

    def filter_mask(self, filter: Union[Filter, dict, Callable[[Optional[Any]], bool]]) -> np.ndarray:
        """bool per internal id: the rows `filter` lets through.  A `Filter` (or its JSON form) goes
        through the metadata store (inverted index for equality on indexed fields, Filter::matches
        otherwise); a plain callable is applied to every live entry's metadata."""
        if isinstance(filter, dict):
            filter = Filter.parse(filter)
        if isinstance(filter, Filter):
            return self._store.filter_bitmask(filter, len(self._vectors))
        keep = np.zeros(len(self._vectors), dtype=np.bool_)
        for i, (_, data) in self._store.entries.items():
            keep[i] = bool(filter(data))
        return keep

    def batch_search_with_filter(self, queries: Sequence[Sequence[float]], k: int,
                                 filter: Union[Filter, dict, Callable[[Optional[Any]], bool]]) -> List[List[SearchResult]]:
        """Collection::batch_search_with_filter, src/collection/batch.rs:108-171; `filter` is a Filter, its
        JSON form, or a callable playing Filter::matches(metadata).  Rows without a metadata entry are
        excluded (search_methods.rs:215-223)."""
        for q in queries:
            self.validate_query(q)
        k = self.clamp_k(k)
        if k == 0:
            return [[] for _ in queries]
        if len(queries) == 0:
            return []
        keep = self.filter_mask(filter)
        ids, dists, counts = self._resident().search(_f32(queries), k, precision=self.precision, filter_mask=keep)
        return [self._enrich(ids[i], dists[i], counts[i]) for i in range(len(queries))]
