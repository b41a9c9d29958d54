"""ctypes binding of libneedle_hip.so (C ABI: include/needle_hip.h).

There is no CPU path: if the shared library is missing this module raises on
first use instead of substituting anything.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEEDLE_HIP_LIBRARY names another build of the same library (kernel development: A/B runs in one job)
LIB_PATH = os.environ.get("NEEDLE_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libneedle_hip.so")

# needle_hip_status
OK, ERR_DIMENSION_MISMATCH, ERR_INVALID_VECTOR, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_HIP, ERR_NOT_TRAINED, \
    ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_RCCL = range(10)

# needle_precision
PRECISION_FP32, PRECISION_BF16, PRECISION_EXACT_SCAN = range(3)
# needle_arith
ARITH_DISTANCE_RS, ARITH_ACCEL_CHUNK4, ARITH_SERVICE, ARITH_AVX2, ARITH_AVX512, ARITH_NEON = range(6)

#: every symbol include/needle_hip.h declares
EXPORTS = (
    "needle_hip_init", "needle_hip_shutdown", "needle_hip_last_error", "needle_hip_version",
    "needle_hip_set_profiling", "needle_hip_get_metrics", "needle_hip_reset_metrics", "needle_hip_set_tuning", "needle_hip_metric_from_str",
    "needle_hip_flat_build", "needle_hip_flat_build_rows", "needle_hip_flat_build_device", "needle_hip_flat_build_half",
    "needle_hip_index_set_valid_mask", "needle_hip_index_append", "needle_hip_index_append_device", "needle_hip_index_swap_remove",
    "needle_hip_index_destroy", "needle_hip_index_len",
    "needle_hip_index_dimensions", "needle_hip_batch_search", "needle_hip_batch_search_device",
    "needle_hip_batch_distance", "needle_hip_kmeans_assign", "needle_hip_merge_topk_device",
    "needle_hip_ivf_build", "needle_hip_ivf_probe", "needle_hip_ivf_list_sizes", "needle_hip_pq_build",
    "needle_hip_sq8_build", "needle_hip_binary_build", "needle_hip_binary_quantize", "needle_hip_ivf_train", "needle_hip_ivf_pq_build", "needle_hip_flat_build_prefix", "needle_hip_adaptive_search", "needle_hip_pq_train", "needle_hip_pq_encode",
    "needle_hip_merge_topk", "needle_hip_normalize", "needle_hip_normalize_device", "needle_hip_batch_top_k",
    "needle_hip_batch_top_k_device", "needle_hip_batch_top_k_device_noalloc",
    "needle_hip_comm_unique_id", "needle_hip_comm_init", "needle_hip_comm_destroy", "needle_hip_comm_rank",
    "needle_hip_comm_world", "needle_hip_sharded_batch_search_device", "needle_hip_shard_slot_bytes",
    "needle_hip_merge_shard_slots_device",
    "needle_hip_distance_matrix", "needle_hip_distance_matrix_device", "needle_hip_multi_query_search",
    "needle_hip_multi_query_search_device",
)

UNIQUE_ID_BYTES = 128


class SearchParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("nprobe", C.c_uint32), ("precision", C.c_int32), ("reserved", C.c_uint32),
                ("filter_bitmask", C.c_void_p)]


class Metrics(C.Structure):
    _fields_ = [("searches", C.c_uint64), ("queries", C.c_uint64), ("fallback_queries", C.c_uint64),
                ("scan_launches", C.c_uint64), ("scan_ms_total", C.c_double), ("last_scan_ms", C.c_double),
                ("last_search_ms", C.c_double), ("refined_queries", C.c_uint64), ("binary_code_scans", C.c_uint64),
                ("list_scans", C.c_uint64), ("list_scan_ms_total", C.c_double), ("compacted_searches", C.c_uint64),
                ("compacted_rows", C.c_uint64)]


class NeedleHipError(RuntimeError):
    """A non-zero needle_hip_status; .status holds the code."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class DimensionMismatch(NeedleHipError, ValueError):
    """NeedleError::DimensionMismatch."""


class InvalidVector(NeedleHipError, ValueError):
    """NeedleError::InvalidVector."""


_lib = None


def _share_torch_hip_runtime() -> None:
    """One process must hold ONE HIP/HSA runtime.  PyTorch's ROCm wheels bundle their own
    libamdhip64.so.7 next to libtorch_hip.so; loading the system copy as well gives the process two
    runtimes and the second one finds no GPU.  If a bundled copy exists, map it first so that this
    library's NEEDED libamdhip64.so.7 binds to it (and a later `import torch` re-uses the same file)."""
    if os.environ.get("NEEDLE_AMD_SYSTEM_HIP") == "1":
        return
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C needle_amd/csrc` "
                          "(or __graft_entry__.build()); needle_amd has no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.needle_hip_init.restype = i32
    L.needle_hip_init.argtypes = [i32, C.POINTER(vp)]
    L.needle_hip_shutdown.restype = None
    L.needle_hip_shutdown.argtypes = [vp]
    L.needle_hip_last_error.restype = C.c_char_p
    L.needle_hip_last_error.argtypes = []
    L.needle_hip_version.restype = i32
    L.needle_hip_set_profiling.restype = i32
    L.needle_hip_set_profiling.argtypes = [vp, i32]
    L.needle_hip_get_metrics.restype = i32
    L.needle_hip_get_metrics.argtypes = [vp, C.POINTER(Metrics)]
    L.needle_hip_reset_metrics.restype = i32
    L.needle_hip_reset_metrics.argtypes = [vp]
    L.needle_hip_set_tuning.restype = i32
    L.needle_hip_set_tuning.argtypes = [vp, C.c_char_p, C.c_double]
    L.needle_hip_metric_from_str.restype = i32
    L.needle_hip_metric_from_str.argtypes = [C.c_char_p]
    L.needle_hip_flat_build.restype = i32
    L.needle_hip_flat_build.argtypes = [vp, vp, u64, u32, i32, i32, vp, C.POINTER(vp)]
    L.needle_hip_flat_build_rows.restype = i32
    L.needle_hip_flat_build_rows.argtypes = [vp, vp, u64, u32, i32, i32, vp, C.POINTER(vp)]
    L.needle_hip_flat_build_device.restype = i32
    L.needle_hip_flat_build_device.argtypes = [vp, vp, u64, u32, i32, i32, vp, C.POINTER(vp)]
    L.needle_hip_index_set_valid_mask.restype = i32
    L.needle_hip_index_set_valid_mask.argtypes = [vp, vp]
    L.needle_hip_index_append.restype = i32
    L.needle_hip_index_append.argtypes = [vp, vp, C.c_uint64, vp]
    L.needle_hip_index_append_device.restype = i32
    L.needle_hip_index_append_device.argtypes = [vp, vp, C.c_uint64, vp]
    L.needle_hip_index_swap_remove.restype = i32
    L.needle_hip_index_swap_remove.argtypes = [vp, C.c_uint64]
    L.needle_hip_index_destroy.restype = None
    L.needle_hip_index_destroy.argtypes = [vp]
    L.needle_hip_index_len.restype = u64
    L.needle_hip_index_len.argtypes = [vp]
    L.needle_hip_index_dimensions.restype = u32
    L.needle_hip_index_dimensions.argtypes = [vp]
    L.needle_hip_batch_search.restype = i32
    L.needle_hip_batch_search.argtypes = [vp, vp, u32, u32, C.POINTER(SearchParams), vp, vp, vp]
    L.needle_hip_batch_search_device.restype = i32
    L.needle_hip_batch_search_device.argtypes = [vp, vp, u32, C.POINTER(SearchParams), vp, vp, vp]
    L.needle_hip_batch_distance.restype = i32
    L.needle_hip_batch_distance.argtypes = [vp, i32, i32, vp, vp, u64, u32, vp]
    L.needle_hip_kmeans_assign.restype = i32
    L.needle_hip_kmeans_assign.argtypes = [vp, i32, vp, u64, vp, u32, u32, vp]
    L.needle_hip_merge_topk_device.restype = i32
    L.needle_hip_merge_topk_device.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, vp, vp, vp]
    L.needle_hip_ivf_build.restype = i32
    L.needle_hip_ivf_build.argtypes = [vp, vp, u64, u32, vp, u32, vp, vp, C.POINTER(vp)]
    L.needle_hip_ivf_probe.restype = i32
    L.needle_hip_ivf_probe.argtypes = [vp, vp, u32, u32, u32, vp, vp]
    L.needle_hip_ivf_list_sizes.restype = i32
    L.needle_hip_ivf_list_sizes.argtypes = [vp, vp]
    L.needle_hip_pq_build.restype = i32
    L.needle_hip_pq_build.argtypes = [vp, vp, u64, u32, u32, vp, vp, C.POINTER(vp)]
    L.needle_hip_sq8_build.restype = i32
    L.needle_hip_sq8_build.argtypes = [vp, vp, u64, u32, vp, vp, vp, C.POINTER(vp)]
    L.needle_hip_ivf_train.restype = i32
    L.needle_hip_ivf_train.argtypes = [vp, vp, u64, u32, u32, u32, C.c_float, u64, vp, C.POINTER(u32)]
    L.needle_hip_flat_build_half.restype = i32
    L.needle_hip_flat_build_half.argtypes = [vp, vp, u64, u32, i32, i32, i32, vp, C.POINTER(vp)]
    L.needle_hip_flat_build_prefix.restype = i32
    L.needle_hip_flat_build_prefix.argtypes = [vp, u32, C.POINTER(vp)]
    L.needle_hip_adaptive_search.restype = i32
    L.needle_hip_adaptive_search.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp, vp, vp, vp]
    L.needle_hip_ivf_pq_build.restype = i32
    L.needle_hip_ivf_pq_build.argtypes = [vp, vp, u64, u32, vp, u32, vp, vp, u32, vp, C.POINTER(vp)]
    L.needle_hip_pq_train.restype = i32
    L.needle_hip_pq_train.argtypes = [vp, vp, u64, u32, u32, u32, u64, vp, vp]
    L.needle_hip_pq_encode.restype = i32
    L.needle_hip_pq_encode.argtypes = [vp, vp, u32, u32, vp, vp, u64, vp]
    L.needle_hip_binary_build.restype = i32
    L.needle_hip_binary_build.argtypes = [vp, vp, u64, u32, vp, vp, C.POINTER(vp)]
    L.needle_hip_binary_quantize.restype = i32
    L.needle_hip_binary_quantize.argtypes = [vp, vp, vp, u64, u32, vp]
    L.needle_hip_merge_topk.restype = i32
    L.needle_hip_merge_topk.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, vp, vp, vp]
    L.needle_hip_normalize.restype = i32
    L.needle_hip_normalize.argtypes = [vp, i32, vp, u64, u32]
    L.needle_hip_normalize_device.restype = i32
    L.needle_hip_normalize_device.argtypes = [vp, i32, vp, u64, u32]
    L.needle_hip_batch_top_k.restype = i32
    L.needle_hip_batch_top_k.argtypes = [vp, vp, u32, u64, u32, vp, vp, vp]
    L.needle_hip_batch_top_k_device.restype = i32
    L.needle_hip_batch_top_k_device.argtypes = [vp, vp, u32, u64, u32, vp, vp, vp]
    for name in ("needle_hip_distance_matrix", "needle_hip_distance_matrix_device"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i32, i32, vp, u32, vp, u64, u32, vp]
    for name in ("needle_hip_multi_query_search", "needle_hip_multi_query_search_device"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i32, i32, vp, u32, vp, u64, u32, u32, vp, vp, vp]
    L.needle_hip_batch_top_k_device_noalloc.restype = i32
    L.needle_hip_batch_top_k_device_noalloc.argtypes = [vp, vp, u32, u64, u32, vp, vp, vp]
    L.needle_hip_comm_unique_id.restype = i32
    L.needle_hip_comm_unique_id.argtypes = [vp]
    L.needle_hip_comm_init.restype = i32
    L.needle_hip_comm_init.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.needle_hip_comm_destroy.restype = None
    L.needle_hip_comm_destroy.argtypes = [vp]
    L.needle_hip_comm_rank.restype = i32
    L.needle_hip_comm_rank.argtypes = [vp]
    L.needle_hip_comm_world.restype = i32
    L.needle_hip_comm_world.argtypes = [vp]
    L.needle_hip_sharded_batch_search_device.restype = i32
    L.needle_hip_sharded_batch_search_device.argtypes = [vp, vp, u64, vp, u32, C.POINTER(SearchParams), vp, vp, vp]
    L.needle_hip_shard_slot_bytes.restype = u64
    L.needle_hip_shard_slot_bytes.argtypes = [u32, u32]
    L.needle_hip_merge_shard_slots_device.restype = i32
    L.needle_hip_merge_shard_slots_device.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp]
    _lib = L
    return L


def check(status: int) -> None:
    if status == OK:
        return
    msg = lib().needle_hip_last_error().decode("utf-8", "replace")
    if status == ERR_DIMENSION_MISMATCH:
        raise DimensionMismatch(status, msg)
    if status == ERR_INVALID_VECTOR:
        raise InvalidVector(status, msg)
    raise NeedleHipError(status, msg)


# ---- orderly shutdown --------------------------------------------------------------------------
# Handles that are still open when the interpreter exits are closed here, indexes before contexts,
# while the HIP runtime is still loaded; finalizers that run later than this must not call into it.
import atexit
import weakref

_live: "weakref.WeakSet" = weakref.WeakSet()
_shutting_down = False


def register_live(obj) -> None:
    _live.add(obj)


def shutting_down() -> bool:
    return _shutting_down


def _close_all() -> None:
    global _shutting_down
    objs = list(_live)
    for o in sorted(objs, key=lambda o: getattr(o, "_close_order", 0)):
        try:
            o.close()
        except Exception:
            pass
    _shutting_down = True


atexit.register(_close_all)
