"""Host-side mirror of the reference's `IvfIndex` / `IvfConfig` (src/indexing/ivf.rs:106-567) over the
device IVF index: same method names, argument meaning and error behaviour.

Search, assignment and training run on the GPU (needle_hip_ivf_build / needle_hip_batch_search /
needle_hip_kmeans_assign / needle_hip_ivf_train).  Training keeps the reference's control flow
(k-means++ seeding, Lloyd, `tolerance` on the largest centroid shift, empty clusters re-seeded) and
arithmetic; the reference draws from `thread_rng`, here the generator is seeded, so centroids are
reproducible but not comparable to any reference run (search parity is defined for given centroids).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .index import PRECISION, Context, _f32, _mask_words, _destroy_handle


@dataclasses.dataclass
class IvfConfig:  # src/indexing/ivf.rs:106-137
    n_clusters: int = 256
    n_probe: int = 8
    max_iterations: int = 100
    tolerance: float = 1e-4
    use_pq: bool = False
    pq_subvectors: int = 8
    pq_bits: int = 8

    @classmethod
    def new(cls, n_clusters: int) -> "IvfConfig":
        return cls(n_clusters=n_clusters)

    def with_nprobe(self, n_probe: int) -> "IvfConfig":
        return dataclasses.replace(self, n_probe=n_probe)

    def with_max_iterations(self, it: int) -> "IvfConfig":
        return dataclasses.replace(self, max_iterations=it)

    def with_tolerance(self, tol: float) -> "IvfConfig":
        return dataclasses.replace(self, tolerance=tol)

    def with_pq(self, subvectors: int) -> "IvfConfig":  # ivf.rs:164-168
        return dataclasses.replace(self, use_pq=True, pq_subvectors=subvectors)


class NotTrained(_lib.NeedleHipError):
    """IvfError::NotTrained."""


class IvfIndex:
    def __init__(self, dimensions: int, config: IvfConfig | None = None, *, device: int = 0,
                 ctx: Context | None = None, seed: int = 1):
        self.dimensions = dimensions
        self.config = config or IvfConfig()
        self.ctx = ctx or Context(device)
        self.pq = None  # ProductQuantizer of the lists when config.use_pq (ivf.rs:321-323)
        self.seed = seed
        self.centroids: np.ndarray | None = None
        self._ids: List[int] = []
        self._rows: List[np.ndarray] = []
        self._h = C.c_void_p()
        self._dirty = True

    # ---- state ---------------------------------------------------------------
    def is_trained(self) -> bool:
        return self.centroids is not None

    def __len__(self) -> int:
        return len(self._ids)

    def is_empty(self) -> bool:
        return not self._ids

    def _check_dim(self, v: np.ndarray) -> None:
        if v.shape[-1] != self.dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dimensions}, got {v.shape[-1]}")

    # ---- train: ivf.rs:299-450 -----------------------------------------------
    def set_centroids(self, centroids) -> None:
        c = _f32(centroids)
        self._check_dim(c)
        self.centroids = c
        self._dirty = True

    def train(self, vectors) -> None:
        v = _f32(vectors)
        if v.shape[0] == 0:
            raise ValueError("Empty training data")  # IvfError::EmptyTrainingData
        self._check_dim(v)
        k = min(self.config.n_clusters, v.shape[0])
        c = np.zeros((k, self.dimensions), dtype=np.float32)
        got = C.c_uint32(0)
        _lib.check(_lib.lib().needle_hip_ivf_train(self.ctx._h, v.ctypes.data, v.shape[0], self.dimensions,
                                                   self.config.n_clusters, self.config.max_iterations,
                                                   self.config.tolerance, self.seed, c.ctypes.data, C.byref(got)))
        c = np.ascontiguousarray(c[: got.value])
        self.centroids = c
        if self.config.use_pq:  # ivf.rs:321-323: the list quantizer is trained on the same rows
            from .quantization import ProductQuantizer

            self.pq = ProductQuantizer.train_device(self.ctx, v, self.config.pq_subvectors, seed=self.seed)
        self._dirty = True

    def set_pq(self, pq) -> None:
        """Use a given (already trained) ProductQuantizer for the lists."""
        if pq.dimensions != self.dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {self.dimensions}, got {pq.dimensions}")
        self.pq = pq
        self._dirty = True

    # ---- insert: ivf.rs:462-506 ----------------------------------------------
    def insert(self, id: int, vector) -> None:
        if not self.is_trained():
            raise NotTrained(_lib.ERR_NOT_TRAINED, "Index not trained")
        v = _f32(vector)
        self._check_dim(v)
        self._ids.append(int(id))
        self._rows.append(v)
        self._dirty = True

    def insert_batch(self, vectors: Sequence[Tuple[int, Sequence[float]]]) -> None:
        for id, v in vectors:
            self.insert(id, v)

    def add(self, ids, vectors) -> None:
        """Bulk insert of a [n, dim] array (same effect as insert() per row)."""
        if not self.is_trained():
            raise NotTrained(_lib.ERR_NOT_TRAINED, "Index not trained")
        v = _f32(vectors)
        self._check_dim(v)
        self._ids.extend(int(i) for i in ids)
        self._rows.extend(v)
        self._dirty = True

    # ---- device index ---------------------------------------------------------
    def close(self) -> None:
        _destroy_handle(self)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _resident(self):
        if not self.is_trained():
            raise NotTrained(_lib.ERR_NOT_TRAINED, "Index not trained")
        if self._dirty or not self._h.value:
            self.close()
            base = np.stack(self._rows) if self._rows else np.zeros((0, self.dimensions), dtype=np.float32)
            self._h = C.c_void_p()
            if self.config.use_pq:
                if self.pq is None:
                    raise NotTrained(_lib.ERR_NOT_TRAINED, "Index not trained")
                books = np.ascontiguousarray(self.pq.codebooks, dtype=np.float32)
                _lib.check(_lib.lib().needle_hip_ivf_pq_build(self.ctx._h, base.ctypes.data if base.shape[0] else None,
                                                              base.shape[0], self.dimensions, self.centroids.ctypes.data,
                                                              self.centroids.shape[0], None, None, self.pq.num_subvectors,
                                                              books.ctypes.data, C.byref(self._h)))
            else:
                _lib.check(_lib.lib().needle_hip_ivf_build(self.ctx._h, base.ctypes.data if base.shape[0] else None,
                                                           base.shape[0], self.dimensions, self.centroids.ctypes.data,
                                                           self.centroids.shape[0], None, None, C.byref(self._h)))
            self._dirty = False
        return self._h

    def cluster_sizes(self) -> np.ndarray:  # cluster_stats sizes, ivf.rs:788-805
        out = np.zeros(self.centroids.shape[0], dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_ivf_list_sizes(self._resident(), out.ctypes.data))
        return out

    def probe(self, queries, n_probe: int | None = None) -> np.ndarray:
        """Coarse step of search(): list ids per query, nearest first (ivf.rs:521-533)."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        self._check_dim(q)
        npb = n_probe or self.config.n_probe
        out = np.full((q.shape[0], npb), np.iinfo(np.uint32).max, dtype=np.uint32)
        cnt = np.zeros(q.shape[0], dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_ivf_probe(self._resident(), q.ctypes.data, q.shape[0], q.shape[1], npb,
                                                   out.ctypes.data, cnt.ctypes.data))
        return out[:, : int(cnt.max())] if q.shape[0] else out

    # ---- search: ivf.rs:509-567 ------------------------------------------------
    def batch_search_arrays(self, queries, k: int, *, precision: str = "fp32", filter_mask=None):
        """(internal row ids[q,k] uint64, dists[q,k], counts[q]); row i is the i-th inserted vector."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        self._check_dim(q)
        h = self._resident()
        nq = q.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fm = _mask_words(filter_mask, len(self._ids))
        sp = _lib.SearchParams(k=k, nprobe=self.config.n_probe, precision=PRECISION[precision], reserved=0,
                               filter_bitmask=fm.ctypes.data if fm is not None else None)
        _lib.check(_lib.lib().needle_hip_batch_search(h, q.ctypes.data if nq else None, nq, q.shape[1], C.byref(sp),
                                                      ids.ctypes.data, dists.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dists[:, :k], counts

    def batch_search(self, queries, k: int) -> List[List[Tuple[int, float]]]:
        ids, dists, counts = self.batch_search_arrays(queries, k)
        return [[(self._ids[int(ids[i, j])], float(dists[i, j])) for j in range(int(counts[i]))]
                for i in range(ids.shape[0])]

    def search(self, query, k: int) -> List[Tuple[int, float]]:
        if not self.is_trained():
            raise NotTrained(_lib.ERR_NOT_TRAINED, "Index not trained")
        return self.batch_search([query], k)[0]
