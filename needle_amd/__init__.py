"""needle_amd -- MI355X-native batched vector search behind Needle's search API.

All compute lives in needle_amd/lib/libneedle_hip.so (hand-written HIP for
gfx950, C ABI in include/needle_hip.h); this package is the host-side mirror of
the reference's operator interface.  There is no CPU fallback.
"""
from . import _lib
from ._lib import DimensionMismatch, InvalidVector, NeedleHipError
from .accelerator import FallbackSearchResult, GpuAccelerator, GpuResidentIndex, TransparentFallbackManager
from .collection import Collection, SearchResult
from .distance import DistanceFunction, DistanceType, normalize, normalized
from .index import Context, FlatIndex, pack_bitmask

__all__ = ["Collection", "Context", "DimensionMismatch", "DistanceFunction", "DistanceType", "FlatIndex",
           "GpuAccelerator", "GpuResidentIndex", "InvalidVector", "NeedleHipError", "SearchResult", "pack_bitmask",
           "_lib", "normalize", "normalized"]
