"""Host-side mirror of the reference's Matryoshka truncation and two-phase search
(src/indexing/quantization.rs:735-943): `MatryoshkaTruncation` (levels, `nearest_level`, `truncate`,
`correct_distance`, `calibrate`, `memory_savings`) and `adaptive_search` over a device-resident index.

Phase 1 (the scan over a prefix of the dimensions) and phase 2 (the re-rank over all of them) run on the GPU
(needle_hip_flat_build_prefix / needle_hip_adaptive_search).  Phase 1 keeps the k * overfetch NEAREST candidates;
the reference's heap evicts its nearest entry once full (quantization.rs:913-918), which its tests never reach.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, List, Sequence

import numpy as np

from . import _lib
from .distance import DistanceFunction
from .index import Context, FlatIndex, _f32

F = np.float32
EPSILON = float(np.finfo(np.float32).eps)


class MatryoshkaTruncation:  # quantization.rs:753-862
    def __init__(self, full_dimensions: int, truncation_levels: Sequence[int]):
        self.full_dimensions = int(full_dimensions)
        levels = sorted((int(d) for d in truncation_levels), reverse=True)          # :768
        self.truncation_levels: List[int] = [d for d in levels if 0 < d < self.full_dimensions]  # :769
        self.variance_ratios: List[float] = []

    def truncate(self, vector, target_dims: int):  # :801-804
        v = np.asarray(vector)
        return v[..., : min(int(target_dims), v.shape[-1])]

    def nearest_level(self, max_dims: int) -> int:  # :825-831: the first (largest) level that fits
        for d in self.truncation_levels:
            if d <= max_dims:
                return d
        return self.full_dimensions

    def correct_distance(self, raw_distance: float, truncated_dims: int) -> float:  # :810-821
        if truncated_dims >= self.full_dimensions or truncated_dims == 0:
            return float(raw_distance)
        if truncated_dims in self.truncation_levels:
            i = self.truncation_levels.index(truncated_dims)
            if i < len(self.variance_ratios):
                return float(F(raw_distance) / F(self.variance_ratios[i]))
        return float(F(raw_distance) * (F(self.full_dimensions) / F(truncated_dims)))

    def memory_savings(self, truncated_dims: int) -> float:  # :844-849
        return 0.0 if truncated_dims == 0 else float(F(self.full_dimensions) / F(truncated_dims))

    @staticmethod
    def _variance(vectors: np.ndarray, dims: int) -> float:  # compute_variance, :851-870: sequential f32 sums
        if vectors.shape[0] == 0 or dims == 0:
            return 0.0
        n = F(vectors.shape[0])
        d = min(dims, vectors.shape[1])
        s = np.zeros(dims, dtype=F)
        sq = np.zeros(dims, dtype=F)
        for v in vectors:
            s[:d] = s[:d] + v[:d]
            sq[:d] = sq[:d] + v[:d] * v[:d]
        total = F(0.0)
        for i in range(dims):
            mean = F(s[i] / n)
            total = F(total + F(F(sq[i] / n) - F(mean * mean)))
        return float(total)

    def calibrate(self, sample_vectors) -> None:  # :779-795
        v = _f32(sample_vectors)
        if v.size == 0 or not self.truncation_levels:
            return
        full_var = self._variance(v, self.full_dimensions)
        if full_var < EPSILON:
            return
        self.variance_ratios = [float(np.clip(F(self._variance(v, d)) / F(full_var), F(0.01), F(1.0)))
                                for d in self.truncation_levels]


@dataclasses.dataclass
class AdaptiveSearchResult:  # quantization.rs:874-882
    id: int
    distance: float
    coarse_distance: float


class AdaptiveIndex:
    """The vectors of `adaptive_search` resident on the GPU, with one prefix index per truncation level used."""

    def __init__(self, ctx: Context, vectors, truncation: MatryoshkaTruncation):
        v = _f32(vectors)
        if v.ndim != 2 or v.shape[1] != truncation.full_dimensions:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"Dimension mismatch: expected {truncation.full_dimensions}, got {v.shape[-1]}")
        self.ctx, self.truncation, self.n = ctx, truncation, v.shape[0]
        self._full = FlatIndex(ctx, v, DistanceFunction.Euclidean)
        self._coarse: Dict[int, C.c_void_p] = {}

    def close(self) -> None:
        ctx_alive = bool(getattr(self.ctx, "_h", None)) and bool(self.ctx._h.value)
        if ctx_alive and not _lib.shutting_down():  # a closed context has taken its indexes along
            for h in self._coarse.values():
                _lib.lib().needle_hip_index_destroy(h)
        self._coarse = {}
        self._full.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _prefix(self, dims: int) -> C.c_void_p:
        if dims not in self._coarse:
            h = C.c_void_p()
            _lib.check(_lib.lib().needle_hip_flat_build_prefix(self._full._h, dims, C.byref(h)))
            self._coarse[dims] = h
        return self._coarse[dims]

    def search_arrays(self, queries, coarse_dims: int, k: int, overfetch_factor: int):
        """(ids[q,k] uint64, full distances[q,k], coarse distances[q,k], counts[q])."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq = q.shape[0]
        dims = self.truncation.nearest_level(coarse_dims)  # quantization.rs:902
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        full = np.full((nq, kk), np.inf, dtype=F)
        coarse = np.full((nq, kk), np.inf, dtype=F)
        counts = np.zeros(nq, dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_adaptive_search(self._full._h, self._prefix(dims), q.ctypes.data if nq else None, nq,
                                                         q.shape[1], k, overfetch_factor, ids.ctypes.data, full.ctypes.data,
                                                         coarse.ctypes.data, counts.ctypes.data))
        return ids[:, :k], full[:, :k], coarse[:, :k], counts

    def adaptive_search(self, query, coarse_dims: int, k: int, overfetch_factor: int) -> List[AdaptiveSearchResult]:
        ids, full, coarse, counts = self.search_arrays(query, coarse_dims, k, overfetch_factor)
        return [AdaptiveSearchResult(int(ids[0, j]), float(full[0, j]), float(coarse[0, j])) for j in range(int(counts[0]))]
