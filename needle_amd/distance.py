"""`DistanceFunction` of the reference (src/distance.rs:43-146), same variants, names and aliases."""
from __future__ import annotations

import enum

from . import _lib


class DistanceFunction(enum.IntEnum):
    Cosine = 0
    CosineNormalized = 1
    Euclidean = 2
    DotProduct = 3
    Manhattan = 4
    Hamming = 5
    Chebyshev = 6

    def __str__(self) -> str:  # Display, src/distance.rs:106-118
        return self.name

    @classmethod
    def from_str(cls, s: str) -> "DistanceFunction":
        """FromStr, src/distance.rs:120-146 (case-insensitive aliases), resolved by the library."""
        m = _lib.lib().needle_hip_metric_from_str(s.encode())
        if m < 0:
            raise ValueError(f"Unknown distance function: '{s}'. Use: cosine, euclidean, dot, manhattan, hamming, "
                             "chebyshev")
        return cls(m)


class DistanceType(enum.IntEnum):
    """`DistanceType` of the operator API (src/experimental/gpu/common.rs)."""
    Cosine = 0
    Euclidean = 1
    DotProduct = 2

    def metric(self) -> DistanceFunction:
        return (DistanceFunction.Cosine, DistanceFunction.Euclidean, DistanceFunction.DotProduct)[int(self)]


def normalized(vector, ctx=None):
    """`normalized` (src/distance.rs:473-477): a unit-length copy of `vector` (a zero vector stays zero), computed on
    the device in the reference's arithmetic (norm summed in order, every component times 1 / norm)."""
    from .index import Context

    own = ctx is None
    c = Context(0) if own else ctx
    try:
        return c.normalize(vector)
    finally:
        if own:
            c.close()


def normalize(vector, ctx=None) -> None:
    """`normalize` (src/distance.rs:426-434): in place on a writable float32 numpy array."""
    vector[...] = normalized(vector, ctx)
