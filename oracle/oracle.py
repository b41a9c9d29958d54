"""ctypes front-end for the CPU oracle (oracle/needle_oracle.c, oracle/needle_hnsw.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the shipped package (needle_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libneedle_oracle.so")

COSINE, COSINE_NORMALIZED, EUCLIDEAN, DOT_PRODUCT, MANHATTAN, HAMMING, CHEBYSHEV = range(7)
ACCEL_COSINE, ACCEL_EUCLIDEAN, ACCEL_DOT = range(3)

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the shared object is missing or stale."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("needle_oracle.c", "needle_hnsw.c"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libneedle_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L: C.CDLL) -> None:
    sz = C.c_size_t
    _declare_hnsw(L)
    for name in ("ndl_dot", "ndl_euclidean_squared", "ndl_euclidean", "ndl_cosine", "ndl_cosine_normalized",
                 "ndl_dot_distance", "ndl_manhattan", "ndl_hamming", "ndl_chebyshev", "ndl_dot_chunk4",
                 "ndl_euclidean_chunk4", "ndl_dot_avx2", "ndl_euclidean_squared_avx2", "ndl_manhattan_avx2",
                 "ndl_dot_avx512", "ndl_euclidean_squared_avx512", "ndl_manhattan_avx512",
                 "ndl_dot_neon", "ndl_euclidean_squared_neon", "ndl_manhattan_neon"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [_f32p, _f32p, sz]
    L.ndl_compute.restype = C.c_float
    L.ndl_compute.argtypes = [C.c_int, _f32p, _f32p, sz]
    L.ndl_cosine_chunk4_precomputed.restype = C.c_float
    L.ndl_cosine_chunk4_precomputed.argtypes = [_f32p, _f32p, sz, C.c_float]
    L.ndl_normalize.restype = None
    L.ndl_normalize.argtypes = [_f32p, sz]
    L.ndl_normalize_avx2.restype = None
    L.ndl_normalize_avx2.argtypes = [_f32p, sz]
    L.ndl_normalize_avx512.restype = None
    L.ndl_normalize_avx512.argtypes = [_f32p, sz]
    L.ndl_normalize_neon.restype = None
    L.ndl_normalize_neon.argtypes = [_f32p, sz]
    L.ndl_metric_from_str.restype = C.c_int
    L.ndl_metric_from_str.argtypes = [C.c_char_p]
    L.ndl_validate_vector.restype = C.c_int
    L.ndl_validate_vector.argtypes = [_f32p, sz, C.POINTER(sz)]
    L.ndl_flat_batch_search.restype = None
    L.ndl_flat_batch_search.argtypes = [_f32p, sz, _f32p, C.c_uint64, sz, sz, C.c_int, _u32p, _u32p, C.c_int,
                                        C.c_int, _u64p, _f32p, _u32p]
    L.ndl_recall.restype = C.c_double
    L.ndl_recall.argtypes = [_u64p, sz, _u64p, sz, sz]
    L.ndl_percentile.restype = C.c_double
    L.ndl_percentile.argtypes = [C.POINTER(C.c_double), sz, C.c_double]
    L.ndl_gen_vectors.restype = None
    L.ndl_gen_vectors.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, sz, _f32p]
    L.ndl_accel_distances.restype = None
    L.ndl_accel_distances.argtypes = [_f32p, _f32p, sz, sz, C.c_int, _f32p]
    L.ndl_accel_batch_distance.restype = None
    L.ndl_accel_batch_distance.argtypes = [_f32p, _f32p, sz, sz, C.c_int, _f32p]
    L.ndl_service_distance.restype = C.c_float
    L.ndl_service_distance.argtypes = [_f32p, _f32p, sz, C.c_int]
    L.ndl_service_distance_matrix.restype = None
    L.ndl_service_distance_matrix.argtypes = [_f32p, sz, _f32p, sz, sz, C.c_int, _f32p]
    L.ndl_top_k.restype = sz
    L.ndl_top_k.argtypes = [_f32p, sz, sz, _u64p, _f32p]
    L.ndl_accel_fused_search.restype = sz
    L.ndl_accel_fused_search.argtypes = [_f32p, _f32p, sz, sz, sz, C.c_int, _u64p, _f32p]
    L.ndl_accel_batch_fused_search.restype = None
    L.ndl_accel_batch_fused_search.argtypes = [_f32p, sz, _f32p, sz, sz, sz, C.c_int, C.c_int, _u64p, _f32p, _u32p]
    L.ndl_accel_streaming_search.restype = sz
    L.ndl_accel_streaming_search.argtypes = [_f32p, _f32p, sz, sz, sz, C.c_int, sz, _u64p, _f32p]
    L.ndl_accel_kmeans_assign.restype = None
    L.ndl_accel_kmeans_assign.argtypes = [_f32p, sz, _f32p, sz, sz, _u64p]
    L.ndl_merge_topk.restype = sz
    L.ndl_merge_topk.argtypes = [_f32p, _u64p, _u32p, sz, sz, sz, _u64p, _f32p]
    L.ndl_ivf_nearest_centroid.restype = C.c_uint32
    L.ndl_ivf_nearest_centroid.argtypes = [_f32p, _f32p, sz, sz]
    L.ndl_ivf_assign.restype = None
    L.ndl_ivf_assign.argtypes = [_f32p, sz, _f32p, sz, sz, C.c_int, _u32p]
    L.ndl_ivf_probe.restype = sz
    L.ndl_ivf_probe.argtypes = [_f32p, _f32p, sz, sz, sz, _u32p]
    L.ndl_ivf_batch_search.restype = None
    L.ndl_ivf_batch_search.argtypes = [_f32p, sz, _f32p, sz, _f32p, sz, _u64p, _u64p, sz, sz, C.c_int, _u8p, _f32p,
                                       sz, C.c_int, _u64p, _f32p, _u32p]
    L.ndl_ivf_kmeans.restype = sz
    L.ndl_ivf_kmeans.argtypes = [_f32p, sz, sz, sz, sz, C.c_float, C.c_uint64, _f32p]
    L.ndl_sq_train.restype = C.c_float
    L.ndl_sq_train.argtypes = [_f32p, sz, sz, _f32p, _f32p, _f32p]
    L.ndl_sq_quantize_batch.restype = None
    L.ndl_sq_quantize_batch.argtypes = [_f32p, sz, sz, _f32p, _f32p, _u8p]
    L.ndl_sq_dequantize.restype = None
    L.ndl_sq_dequantize.argtypes = [_u8p, sz, _f32p, _f32p, _f32p]
    L.ndl_sq_asymmetric.restype = C.c_float
    L.ndl_sq_asymmetric.argtypes = [_f32p, _u8p, sz, _f32p, _f32p]
    L.ndl_sq_distance_squared.restype = C.c_float
    L.ndl_sq_distance_squared.argtypes = [_u8p, _u8p, sz, C.c_float]
    L.ndl_pq_train.restype = None
    L.ndl_pq_train.argtypes = [_f32p, sz, sz, sz, C.c_uint64, _f32p, _u32p]
    L.ndl_pq_encode_batch.restype = None
    L.ndl_pq_encode_batch.argtypes = [_f32p, sz, sz, sz, _f32p, _u32p, _u8p]
    L.ndl_pq_decode.restype = None
    L.ndl_pq_decode.argtypes = [_u8p, sz, sz, _f32p, _f32p]
    L.ndl_pq_asymmetric.restype = C.c_float
    L.ndl_pq_asymmetric.argtypes = [_f32p, _u8p, sz, sz, _f32p]
    L.ndl_pq_distance_table.restype = None
    L.ndl_pq_distance_table.argtypes = [_f32p, sz, sz, _f32p, _f32p]
    L.ndl_pq_distance_with_table.restype = C.c_float
    L.ndl_pq_distance_with_table.argtypes = [_f32p, _u8p, sz]
    L.ndl_sq_batch_search.restype = None
    L.ndl_sq_batch_search.argtypes = [_f32p, sz, sz, _u8p, C.c_uint64, _f32p, _f32p, sz, _u32p, C.c_int, _u64p,
                                      _f32p, _u32p]
    _u16p = C.POINTER(C.c_uint16)
    L.ndl_half_to_f32.restype = None
    L.ndl_half_to_f32.argtypes = [C.c_int, _u16p, sz, _f32p]
    L.ndl_half_from_f32.restype = None
    L.ndl_half_from_f32.argtypes = [C.c_int, C.c_int, _f32p, sz, _u16p]
    L.ndl_adaptive_search.restype = sz
    L.ndl_adaptive_search.argtypes = [_f32p, _f32p, C.c_uint64, sz, sz, sz, sz, C.c_int, _u64p, _f32p, _f32p]
    L.ndl_bq_train.restype = None
    L.ndl_bq_train.argtypes = [_f32p, sz, sz, _f32p]
    L.ndl_bq_quantize.restype = None
    L.ndl_bq_quantize.argtypes = [_f32p, sz, _f32p, _u8p]
    L.ndl_bq_hamming.restype = C.c_uint32
    L.ndl_bq_hamming.argtypes = [_u8p, _u8p, sz]
    L.ndl_bq_batch_search.restype = None
    L.ndl_bq_batch_search.argtypes = [_f32p, sz, sz, _u8p, C.c_uint64, _f32p, sz, _u32p, C.c_int, _u64p, _f32p, _u32p]
    L.ndl_pq_batch_search.restype = None
    L.ndl_pq_batch_search.argtypes = [_f32p, sz, sz, _u8p, C.c_uint64, sz, _f32p, sz, _u32p, C.c_int, _u64p, _f32p,
                                      _u32p]


# ----------------------------------------------------------------------------
# numpy helpers
# ----------------------------------------------------------------------------
def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def _mask_ptr(mask, n):
    """bool[n] or packed uint32 bitmask -> (keepalive, pointer-or-None)."""
    if mask is None:
        return None, None
    m = np.asarray(mask)
    if m.dtype == np.bool_:
        m = pack_bitmask(m)
    m = np.ascontiguousarray(m, dtype=np.uint32)
    assert m.size >= (n + 31) // 32
    return m, _p(m, _u32p)


def pack_bitmask(bits) -> np.ndarray:
    """bool[n] -> uint32[ceil(n/32)], bit i of word i>>5 at position i&31."""
    b = np.asarray(bits, dtype=np.bool_)
    n = b.size
    pad = (-n) % 32
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=np.bool_)])
    return np.packbits(b.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy()


def cpu_threads() -> int:
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


# ----------------------------------------------------------------------------
# scalar distances
# ----------------------------------------------------------------------------
def compute(metric: int, a, b) -> float:
    a, b = _f32(a), _f32(b)
    if a.size != b.size:
        raise ValueError(f"Dimension mismatch: expected {a.size}, got {b.size}")
    return float(lib().ndl_compute(metric, _p(a, _f32p), _p(b, _f32p), a.size))


def _pair(name):
    def f(a, b) -> float:
        a, b = _f32(a), _f32(b)
        if a.size != b.size:
            raise ValueError(f"Dimension mismatch: expected {a.size}, got {b.size}")
        return float(getattr(lib(), name)(_p(a, _f32p), _p(b, _f32p), a.size))
    f.__name__ = name
    return f


dot = _pair("ndl_dot")
euclidean = _pair("ndl_euclidean")
euclidean_squared = _pair("ndl_euclidean_squared")
cosine = _pair("ndl_cosine")
cosine_normalized = _pair("ndl_cosine_normalized")
dot_distance = _pair("ndl_dot_distance")
manhattan = _pair("ndl_manhattan")
hamming = _pair("ndl_hamming")
chebyshev = _pair("ndl_chebyshev")
dot_chunk4 = _pair("ndl_dot_chunk4")
euclidean_chunk4 = _pair("ndl_euclidean_chunk4")
# the arithmetic of a reference build with AVX2 enabled (src/distance.rs:481-591); AVX2_ORDER or-ed into a metric selects
# it in compute() and flat_batch_search()
AVX2_ORDER = 0x100
dot_avx2 = _pair("ndl_dot_avx2")
euclidean_squared_avx2 = _pair("ndl_euclidean_squared_avx2")
manhattan_avx2 = _pair("ndl_manhattan_avx2")
AVX512_ORDER = 0x200  # the same for a build with AVX-512 enabled (src/distance.rs:613-715)
dot_avx512 = _pair("ndl_dot_avx512")
euclidean_squared_avx512 = _pair("ndl_euclidean_squared_avx512")
manhattan_avx512 = _pair("ndl_manhattan_avx512")
NEON_ORDER = 0x400  # the aarch64 build (src/distance.rs:741-834)
dot_neon = _pair("ndl_dot_neon")
euclidean_squared_neon = _pair("ndl_euclidean_squared_neon")
manhattan_neon = _pair("ndl_manhattan_neon")


def cosine_chunk4_precomputed(a, b, a_norm: float) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().ndl_cosine_chunk4_precomputed(_p(a, _f32p), _p(b, _f32p), a.size, C.c_float(a_norm)))


def normalized(v, avx2: bool = False, avx512: bool = False, neon: bool = False) -> np.ndarray:
    out = _f32(v).copy()
    f = lib().ndl_normalize_neon if neon else (
        lib().ndl_normalize_avx512 if avx512 else (lib().ndl_normalize_avx2 if avx2 else lib().ndl_normalize))
    f(_p(out, _f32p), out.size)
    return out


def normalize_rows(m) -> np.ndarray:
    out = _f32(m).copy()
    L = lib()
    for i in range(out.shape[0]):
        row = out[i]
        L.ndl_normalize(_p(row, _f32p), row.size)
    return out


def metric_from_str(s: str) -> int:
    return int(lib().ndl_metric_from_str(s.encode()))


def validate_vector(v):
    v = _f32(v)
    bad = C.c_size_t(0)
    rc = lib().ndl_validate_vector(_p(v, _f32p), v.size, C.byref(bad))
    return rc, int(bad.value)


# ----------------------------------------------------------------------------
# flat search / ground truth
# ----------------------------------------------------------------------------
def flat_batch_search(queries, base, k: int, metric: int, valid=None, filt=None, evict_nearest: bool = False,
                      threads: int | None = None):
    """Returns (ids[Q,k] uint64, dists[Q,k] float32, counts[Q] uint32)."""
    q, b = _f32(queries), _f32(base)
    nq, dim = q.shape
    n = b.shape[0]
    assert b.ndim == 2 and (n == 0 or b.shape[1] == dim)
    kk = max(k, 1)
    ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
    dists = np.full((nq, kk), np.inf, dtype=np.float32)
    counts = np.zeros(nq, dtype=np.uint32)
    vk, vp = _mask_ptr(valid, n)
    fk, fp = _mask_ptr(filt, n)
    if nq and k:
        lib().ndl_flat_batch_search(_p(q, _f32p), nq, _p(b, _f32p), n, dim, k, metric, vp, fp, int(evict_nearest),
                                    threads or cpu_threads(), _p(ids, _u64p), _p(dists, _f32p), _p(counts, _u32p))
    return ids[:, :k], dists[:, :k], counts


def recall(ann_ids, gt_ids, k: int) -> float:
    a = np.ascontiguousarray(ann_ids, dtype=np.uint64)
    g = np.ascontiguousarray(gt_ids, dtype=np.uint64)
    return float(lib().ndl_recall(_p(a, _u64p), a.size, _p(g, _u64p), g.size, k))


def percentile(sorted_vals, p: float) -> float:
    s = np.ascontiguousarray(sorted_vals, dtype=np.float64)
    return float(lib().ndl_percentile(s.ctypes.data_as(C.POINTER(C.c_double)), s.size, p))


def gen_vectors(seed: int, first: int, count: int, dim: int) -> np.ndarray:
    out = np.empty((count, dim), dtype=np.float32)
    lib().ndl_gen_vectors(seed, first, count, dim, _p(out, _f32p))
    return out


# ----------------------------------------------------------------------------
# GpuAccelerator CPU arm
# ----------------------------------------------------------------------------
def accel_batch_distance(query, base, dtype: int) -> np.ndarray:
    q, b = _f32(query), _f32(base)
    out = np.empty(b.shape[0], dtype=np.float32)
    lib().ndl_accel_batch_distance(_p(q, _f32p), _p(b, _f32p), b.shape[0], q.size, dtype, _p(out, _f32p))
    return out


def service_distance(a, b, dtype: int) -> float:
    """gpu_kernels.rs:345-358 (the compute service's own cosine / euclidean / dot product)."""
    x, y = _f32(a), _f32(b)
    return float(lib().ndl_service_distance(_p(x, _f32p), _p(y, _f32p), x.size, dtype))


def service_distance_matrix(queries, base, dtype: int) -> np.ndarray:
    """distance_matrix_cosine / _euclidean, gpu_kernels.rs:223-265: [queries][vectors]."""
    q, b = _f32(queries), _f32(base)
    out = np.empty((q.shape[0], b.shape[0]), dtype=np.float32)
    if out.size:
        lib().ndl_service_distance_matrix(_p(q, _f32p), q.shape[0], _p(b, _f32p), b.shape[0], q.shape[1], dtype,
                                          _p(out, _f32p))
    return out


def top_k(scores, k: int):
    s = _f32(scores)
    kk = max(min(k, s.size), 1)
    idx = np.empty(kk, dtype=np.uint64)
    val = np.empty(kk, dtype=np.float32)
    m = lib().ndl_top_k(_p(s, _f32p), s.size, k, _p(idx, _u64p), _p(val, _f32p))
    return idx[:m], val[:m]


def accel_fused_search(query, base, k: int, dtype: int):
    q, b = _f32(query), _f32(base)
    kk = max(min(k, b.shape[0]), 1)
    idx = np.empty(kk, dtype=np.uint64)
    val = np.empty(kk, dtype=np.float32)
    m = lib().ndl_accel_fused_search(_p(q, _f32p), _p(b, _f32p), b.shape[0], q.size, min(k, b.shape[0]), dtype,
                                     _p(idx, _u64p), _p(val, _f32p))
    return idx[:m], val[:m]


def accel_batch_fused_search(queries, base, k: int, dtype: int, threads: int | None = None):
    q, b = _f32(queries), _f32(base)
    nq, dim = q.shape
    n = b.shape[0]
    k = min(k, n)
    kk = max(k, 1)
    idx = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
    val = np.full((nq, kk), np.inf, dtype=np.float32)
    counts = np.zeros(nq, dtype=np.uint32)
    if nq and n and k:
        lib().ndl_accel_batch_fused_search(_p(q, _f32p), nq, _p(b, _f32p), n, dim, k, dtype, threads or cpu_threads(),
                                           _p(idx, _u64p), _p(val, _f32p), _p(counts, _u32p))
    return idx[:, :k], val[:, :k], counts


def accel_streaming_search(query, base, k: int, dtype: int, chunk: int):
    q, b = _f32(query), _f32(base)
    kk = max(k, 1)
    idx = np.empty(kk, dtype=np.uint64)
    val = np.empty(kk, dtype=np.float32)
    m = lib().ndl_accel_streaming_search(_p(q, _f32p), _p(b, _f32p), b.shape[0], q.size, k, dtype, chunk,
                                         _p(idx, _u64p), _p(val, _f32p))
    return idx[:m], val[:m]


def accel_kmeans_assign(vectors, centroids) -> np.ndarray:
    v, c = _f32(vectors), _f32(centroids)
    out = np.empty(v.shape[0], dtype=np.uint64)
    lib().ndl_accel_kmeans_assign(_p(v, _f32p), v.shape[0], _p(c, _f32p), c.shape[0], v.shape[1], _p(out, _u64p))
    return out


def merge_topk(dists, ids, counts, k: int):
    """dists/ids: [lists, per]; counts: [lists]."""
    d = _f32(dists)
    i = np.ascontiguousarray(ids, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    lists, per = d.shape
    kk = max(k, 1)
    oi = np.empty(kk, dtype=np.uint64)
    od = np.empty(kk, dtype=np.float32)
    m = lib().ndl_merge_topk(_p(d, _f32p), _p(i, _u64p), _p(c, _u32p), lists, per, k, _p(oi, _u64p), _p(od, _f32p))
    return oi[:m], od[:m]


# ----------------------------------------------------------------------------
# IVF
# ----------------------------------------------------------------------------
def ivf_assign(base, centroids, threads: int | None = None) -> np.ndarray:
    b, c = _f32(base), _f32(centroids)
    out = np.empty(b.shape[0], dtype=np.uint32)
    lib().ndl_ivf_assign(_p(b, _f32p), b.shape[0], _p(c, _f32p), c.shape[0], b.shape[1], threads or cpu_threads(),
                         _p(out, _u32p))
    return out


def ivf_lists_from_assignment(assign, nlist: int):
    """CSR inverted lists in insertion (row) order: (offsets[nlist+1], order[N])."""
    a = np.asarray(assign, dtype=np.int64)
    order = np.argsort(a, kind="stable").astype(np.uint64)
    counts = np.bincount(a, minlength=nlist)
    offsets = np.zeros(nlist + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    return offsets, order


def ivf_probe(query, centroids, nprobe: int) -> np.ndarray:
    q, c = _f32(query), _f32(centroids)
    out = np.empty(max(min(nprobe, c.shape[0]), 1), dtype=np.uint32)
    m = lib().ndl_ivf_probe(_p(q, _f32p), _p(c, _f32p), c.shape[0], q.size, nprobe, _p(out, _u32p))
    return out[:m]


def ivf_batch_search(queries, base, centroids, offsets, order, nprobe: int, k: int, evict_nearest: bool = False,
                     pq_codes=None, pq_codebooks=None, pq_m: int = 0, threads: int | None = None):
    q, c = _f32(queries), _f32(centroids)
    b = _f32(base) if base is not None else None
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    od = np.ascontiguousarray(order, dtype=np.uint64)
    nq, dim = q.shape
    kk = max(k, 1)
    ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
    dists = np.full((nq, kk), np.inf, dtype=np.float32)
    counts = np.zeros(nq, dtype=np.uint32)
    codes = np.ascontiguousarray(pq_codes, dtype=np.uint8) if pq_codes is not None else None
    cbs = _f32(pq_codebooks) if pq_codebooks is not None else None
    if nq and k:
        lib().ndl_ivf_batch_search(_p(q, _f32p), nq, _p(b, _f32p) if b is not None else None, dim, _p(c, _f32p),
                                   c.shape[0], _p(off, _u64p), _p(od, _u64p), nprobe, k, int(evict_nearest),
                                   _p(codes, _u8p) if codes is not None else None,
                                   _p(cbs, _f32p) if cbs is not None else None, pq_m, threads or cpu_threads(),
                                   _p(ids, _u64p), _p(dists, _f32p), _p(counts, _u32p))
    return ids[:, :k], dists[:, :k], counts


def ivf_kmeans(vectors, nlist: int, max_iter: int = 100, tol: float = 1e-4, seed: int = 1) -> np.ndarray:
    v = _f32(vectors)
    n, dim = v.shape
    cents = np.zeros((min(nlist, n), dim), dtype=np.float32)
    k = lib().ndl_ivf_kmeans(_p(v, _f32p), n, dim, nlist, max_iter, C.c_float(tol), seed, _p(cents, _f32p))
    return cents[:k]


# ----------------------------------------------------------------------------
# quantizers
# ----------------------------------------------------------------------------
class ScalarQuantizer:
    """ScalarQuantizer of src/indexing/quantization.rs:135-286."""

    def __init__(self, min_vals, max_vals, scale, avg_scale):
        self.min_vals, self.max_vals, self.scale, self.avg_scale = min_vals, max_vals, scale, avg_scale
        self.dimensions = min_vals.size

    @classmethod
    def train(cls, vectors):
        v = _f32(vectors)
        n, dim = v.shape
        mn = np.empty(dim, dtype=np.float32)
        mx = np.empty(dim, dtype=np.float32)
        sc = np.empty(dim, dtype=np.float32)
        avg = lib().ndl_sq_train(_p(v, _f32p), n, dim, _p(mn, _f32p), _p(mx, _f32p), _p(sc, _f32p))
        return cls(mn, mx, sc, float(avg))

    def quantize(self, vectors) -> np.ndarray:
        v = _f32(vectors)
        single = v.ndim == 1
        v2 = v.reshape(1, -1) if single else v
        out = np.empty(v2.shape, dtype=np.uint8)
        lib().ndl_sq_quantize_batch(_p(v2, _f32p), v2.shape[0], v2.shape[1], _p(self.min_vals, _f32p),
                                    _p(self.scale, _f32p), _p(out, _u8p))
        return out[0] if single else out

    def dequantize(self, codes) -> np.ndarray:
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty(c.size, dtype=np.float32)
        lib().ndl_sq_dequantize(_p(c, _u8p), c.size, _p(self.min_vals, _f32p), _p(self.scale, _f32p), _p(out, _f32p))
        return out

    def asymmetric_distance(self, query, codes) -> float:
        q = _f32(query)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return float(lib().ndl_sq_asymmetric(_p(q, _f32p), _p(c, _u8p), q.size, _p(self.min_vals, _f32p),
                                             _p(self.scale, _f32p)))

    def distance_squared(self, a, b) -> float:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return float(lib().ndl_sq_distance_squared(_p(a, _u8p), _p(b, _u8p), a.size, C.c_float(self.avg_scale)))

    def batch_search(self, queries, codes, k: int, valid=None, threads: int | None = None):
        q = _f32(queries)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        nq, dim = q.shape
        n = c.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        vk, vp = _mask_ptr(valid, n)
        lib().ndl_sq_batch_search(_p(q, _f32p), nq, dim, _p(c, _u8p), n, _p(self.min_vals, _f32p),
                                  _p(self.scale, _f32p), k, vp, threads or cpu_threads(), _p(ids, _u64p),
                                  _p(dists, _f32p), _p(counts, _u32p))
        return ids[:, :k], dists[:, :k], counts


class ProductQuantizer:
    """ProductQuantizer of src/indexing/quantization.rs:317-504 (seeded training)."""

    def __init__(self, codebooks, ncent, dim, m):
        self.codebooks, self.ncent, self.dim, self.m = codebooks, ncent, dim, m
        self.dsub = dim // m

    @classmethod
    def train(cls, vectors, m: int, seed: int = 1):
        v = _f32(vectors)
        n, dim = v.shape
        cb = np.zeros((m, 256, dim // m), dtype=np.float32)
        nc = np.zeros(m, dtype=np.uint32)
        lib().ndl_pq_train(_p(v, _f32p), n, dim, m, seed, _p(cb, _f32p), _p(nc, _u32p))
        return cls(cb, nc, dim, m)

    def encode(self, vectors) -> np.ndarray:
        v = _f32(vectors)
        single = v.ndim == 1
        v2 = v.reshape(1, -1) if single else v
        out = np.empty((v2.shape[0], self.m), dtype=np.uint8)
        lib().ndl_pq_encode_batch(_p(v2, _f32p), v2.shape[0], self.dim, self.m, _p(self.codebooks, _f32p),
                                  _p(self.ncent, _u32p), _p(out, _u8p))
        return out[0] if single else out

    def decode(self, codes) -> np.ndarray:
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty(self.m * self.dsub, dtype=np.float32)
        lib().ndl_pq_decode(_p(c, _u8p), self.dim, self.m, _p(self.codebooks, _f32p), _p(out, _f32p))
        return out

    def asymmetric_distance(self, query, codes) -> float:
        q = _f32(query)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return float(lib().ndl_pq_asymmetric(_p(q, _f32p), _p(c, _u8p), self.dim, self.m, _p(self.codebooks, _f32p)))

    def compute_distance_table(self, query) -> np.ndarray:
        q = _f32(query)
        t = np.empty((self.m, 256), dtype=np.float32)
        lib().ndl_pq_distance_table(_p(q, _f32p), self.dim, self.m, _p(self.codebooks, _f32p), _p(t, _f32p))
        return t

    def distance_with_table(self, table, codes) -> float:
        t = _f32(table)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return float(lib().ndl_pq_distance_with_table(_p(t, _f32p), _p(c, _u8p), self.m))

    def batch_search(self, queries, codes, k: int, valid=None, threads: int | None = None):
        q = _f32(queries)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        nq, dim = q.shape
        n = c.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        vk, vp = _mask_ptr(valid, n)
        lib().ndl_pq_batch_search(_p(q, _f32p), nq, dim, _p(c, _u8p), n, self.m, _p(self.codebooks, _f32p), k, vp,
                                  threads or cpu_threads(), _p(ids, _u64p), _p(dists, _f32p), _p(counts, _u32p))
        return ids[:, :k], dists[:, :k], counts


class BinaryQuantizer:
    """BinaryQuantizer of src/indexing/quantization.rs:536-594 and its search call site
    (src/indexing/compression_advisor.rs:658-671)."""

    def __init__(self, thresholds):
        self.thresholds = _f32(thresholds)
        self.dimensions = self.thresholds.size

    @classmethod
    def train(cls, vectors):
        v = _f32(vectors)
        if v.size == 0:
            return cls(np.zeros(0, dtype=np.float32))
        t = np.empty(v.shape[1], dtype=np.float32)
        lib().ndl_bq_train(_p(v, _f32p), v.shape[0], v.shape[1], _p(t, _f32p))
        return cls(t)

    def quantize(self, vectors) -> np.ndarray:
        v = _f32(vectors)
        single = v.ndim == 1
        v2 = v.reshape(1, -1) if single else v
        out = np.zeros((v2.shape[0], (self.dimensions + 7) // 8), dtype=np.uint8)
        for i in range(v2.shape[0]):
            lib().ndl_bq_quantize(_p(v2[i], _f32p), self.dimensions, _p(self.thresholds, _f32p),
                                  out[i].ctypes.data_as(_u8p))
        return out[0] if single else out

    @staticmethod
    def hamming_distance(a, b) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return int(lib().ndl_bq_hamming(_p(a, _u8p), _p(b, _u8p), min(a.size, b.size)))

    def batch_search(self, queries, codes, k: int, valid=None, threads: int | None = None):
        q = _f32(queries)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        nq, dim = q.shape
        n = c.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        vk, vp = _mask_ptr(valid, n)
        lib().ndl_bq_batch_search(_p(q, _f32p), nq, dim, _p(c, _u8p), n, _p(self.thresholds, _f32p), k, vp,
                                  threads or cpu_threads(), _p(ids, _u64p), _p(dists, _f32p), _p(counts, _u32p))
        return ids[:, :k], dists[:, :k], counts


def adaptive_search(query, vectors, coarse_dims: int, k: int, overfetch: int, evict_nearest: bool = False):
    """adaptive_search of src/indexing/quantization.rs:890-943 -> (ids, full distances, coarse distances)."""
    q, v = _f32(query), _f32(vectors)
    kk = max(k, 1)
    ids = np.empty(kk, dtype=np.uint64)
    full = np.empty(kk, dtype=np.float32)
    coarse = np.empty(kk, dtype=np.float32)
    m = lib().ndl_adaptive_search(_p(q, _f32p), _p(v, _f32p), v.shape[0], v.shape[1], coarse_dims, k, overfetch,
                                  int(evict_nearest), _p(ids, _u64p), _p(full, _f32p), _p(coarse, _f32p))
    return ids[:m], full[:m], coarse[:m]


HALF_F16, HALF_BF16 = 0, 1


def half_to_f32(kind: int, bits) -> np.ndarray:
    """F16::to_f32 / Bf16::to_f32 of src/indexing/float16.rs over an array of raw 16-bit patterns."""
    b = np.ascontiguousarray(bits, dtype=np.uint16)
    out = np.empty(b.shape, dtype=np.float32)
    lib().ndl_half_to_f32(kind, b.ctypes.data_as(C.POINTER(C.c_uint16)), b.size, _p(out, _f32p))
    return out


def half_from_f32(kind: int, values, rounded: bool = False) -> np.ndarray:
    """F16::from_f32 (truncating), Bf16::from_f32 (truncating) or Bf16::from_f32_round."""
    v = _f32(values)
    out = np.empty(v.shape, dtype=np.uint16)
    lib().ndl_half_from_f32(kind, int(rounded), _p(v, _f32p), v.size, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def _declare_hnsw(L: C.CDLL) -> None:
    sz, vp = C.c_size_t, C.c_void_p
    L.ndl_hnsw_new.restype = vp
    L.ndl_hnsw_new.argtypes = [_f32p, sz, sz, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.ndl_hnsw_free.restype = None
    L.ndl_hnsw_free.argtypes = [vp]
    L.ndl_hnsw_insert.restype = C.c_int
    L.ndl_hnsw_insert.argtypes = [vp, C.c_uint32]
    L.ndl_hnsw_insert_range.restype = None
    L.ndl_hnsw_insert_range.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ndl_hnsw_delete.restype = C.c_int
    L.ndl_hnsw_delete.argtypes = [vp, C.c_uint32]
    L.ndl_hnsw_set_ef_search.restype = None
    L.ndl_hnsw_set_ef_search.argtypes = [vp, C.c_uint32]
    L.ndl_hnsw_len.restype = sz
    L.ndl_hnsw_len.argtypes = [vp]
    L.ndl_hnsw_set_fast_build.restype = None
    L.ndl_hnsw_set_fast_build.argtypes = [vp, C.c_int]
    L.ndl_hnsw_entry_level.restype = C.c_uint32
    L.ndl_hnsw_entry_level.argtypes = [vp]
    L.ndl_hnsw_degree.restype = sz
    L.ndl_hnsw_degree.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ndl_hnsw_search.restype = sz
    L.ndl_hnsw_search.argtypes = [vp, _f32p, sz, sz, _u64p, _f32p, C.POINTER(sz)]
    L.ndl_hnsw_batch_search.restype = None
    L.ndl_hnsw_batch_search.argtypes = [vp, _f32p, sz, sz, _u64p, _f32p, _u32p, C.c_int]


class Hnsw:
    """HnswIndex of src/indexing/hnsw.rs over a fixed row-major vectors array (ids are row numbers), with the
    reference's default configuration (:249-259) unless overridden.  Levels are drawn from a seeded generator (the
    reference's are unseeded), so a graph is reproducible here and unpinned against the reference."""

    def __init__(self, vectors, metric: int, m: int = 16, m_max_0: int | None = None, ef_construction: int = 200,
                 ef_search: int = 50, seed: int = 42, fast_build: bool = True):
        self.vectors = _f32(vectors).reshape(len(vectors), -1) if len(vectors) else np.zeros((0, 1), np.float32)
        self.ef_search = ef_search
        self._h = lib().ndl_hnsw_new(_p(self.vectors, _f32p), self.vectors.shape[0], self.vectors.shape[1], metric, m,
                                     2 * m if m_max_0 is None else m_max_0, ef_construction, ef_search, seed)
        if not self._h:
            raise ValueError("invalid HNSW configuration")  # HnswConfig::validate, :326-357
        # the build scores the unvisited neighbours of a candidate together (interleaved sums: the same bits, several
        # times the speed); False: row by row, as the searches always do
        lib().ndl_hnsw_set_fast_build(self._h, int(fast_build))

    def insert(self, row: int) -> None:
        if lib().ndl_hnsw_insert(self._h, row) != 0:
            raise IndexError(row)

    def insert_range(self, first: int, count: int) -> None:
        assert 0 <= first and first + count <= self.vectors.shape[0]
        lib().ndl_hnsw_insert_range(self._h, first, count)

    def delete(self, row: int) -> bool:
        return bool(lib().ndl_hnsw_delete(self._h, row))

    def set_ef_search(self, ef: int) -> None:
        """set_ef_search (:566-568)."""
        self.ef_search = ef
        lib().ndl_hnsw_set_ef_search(self._h, ef)

    def __len__(self) -> int:
        return lib().ndl_hnsw_len(self._h)

    @property
    def entry_level(self) -> int:
        return lib().ndl_hnsw_entry_level(self._h)

    def degree(self, row: int, layer: int = 0) -> int:
        return lib().ndl_hnsw_degree(self._h, row, layer)

    def search(self, query, k: int, ef: int | None = None):
        """search / search_with_ef (:701-771): (ids, distances, visited_nodes)."""
        q = _f32(query)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        dists = np.empty(max(k, 1), dtype=np.float32)
        vis = C.c_size_t(0)
        n = lib().ndl_hnsw_search(self._h, _p(q, _f32p), k, self.ef_search if ef is None else ef, _p(ids, _u64p),
                                  _p(dists, _f32p), C.byref(vis))
        return ids[:n], dists[:n], vis.value

    def batch_search(self, queries, k: int, threads: int = 1):
        """Collection::batch_search (src/collection/batch.rs:32-52): [nq, k] ids and distances, counts."""
        q = _f32(queries)
        nq = q.shape[0]
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dists = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        lib().ndl_hnsw_batch_search(self._h, _p(q, _f32p), nq, k, _p(ids, _u64p), _p(dists, _f32p), _p(counts, _u32p),
                                    threads)
        return ids, dists, counts

    def close(self) -> None:
        if self._h:
            lib().ndl_hnsw_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
