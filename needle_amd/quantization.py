"""Host-side mirrors of the reference's quantizers (src/indexing/quantization.rs) with the scans on
the GPU: `ScalarQuantizer` (:135-286) and `ProductQuantizer` (:317-504).

The quantizer objects hold the trained parameters and keep the reference's method names; the code
matrices live in device indexes (`Sq8Index`, `PqIndex`) whose `search` returns the k rows nearest by
the quantizer's asymmetric distance, bit-equal to the reference arithmetic.  Training and encoding run
on the host here (numpy, same f32 operations as the reference); the reference trains its codebooks
from `thread_rng`, so only the seeded control flow is mirrored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .index import PRECISION, Context, _f32, _mask_words, _destroy_handle

F = np.float32


class ScalarQuantizer:
    """Per-dimension affine u8 quantizer (quantization.rs:135-286)."""

    def __init__(self, min_vals, max_vals, scale):
        self.min_vals = _f32(min_vals)
        self.max_vals = _f32(max_vals)
        self.scale = _f32(scale)
        self.dimensions = self.min_vals.size

    @classmethod
    def train(cls, vectors) -> "ScalarQuantizer":  # quantization.rs:154-211
        v = _f32(vectors)
        if v.shape[0] == 0:
            return cls(np.zeros(0, F), np.zeros(0, F), np.zeros(0, F))
        if not np.isfinite(v).all():
            raise ValueError("Training vector contains non-finite value")
        mn, mx = v.min(axis=0), v.max(axis=0)
        rng = (mx - mn).astype(F)
        scale = np.where(rng > F(1e-10), F(255.0) / rng, F(1.0)).astype(F)
        return cls(mn, mx, scale)

    def quantize(self, vectors) -> np.ndarray:  # quantization.rs:232-238: clamp, then truncating cast
        v = _f32(vectors)
        x = ((v - self.min_vals) * self.scale).astype(F)
        x = np.where(np.isnan(x), F(0), np.clip(x, F(0), F(255)))
        return x.astype(np.uint8)

    def dequantize(self, codes) -> np.ndarray:  # quantization.rs:254-260
        c = np.asarray(codes, dtype=np.uint8)
        return ((c.astype(F) / self.scale).astype(F) + self.min_vals).astype(F)


class ProductQuantizer:
    """256-centroid product quantizer (quantization.rs:317-504)."""

    def __init__(self, codebooks):
        self.codebooks = _f32(codebooks)  # [m][256][dsub]
        self.num_subvectors, _, self.subvector_dim = self.codebooks.shape
        self.dimensions = self.num_subvectors * self.subvector_dim

    @classmethod
    def train(cls, vectors, num_subvectors: int, seed: int = 1, iterations: int = 20) -> "ProductQuantizer":
        """quantization.rs:330-364 / kmeans :597-711 (k-means++ style seeding, 20 Lloyd iterations,
        empty clusters re-seeded), on raw sub-vectors, with a seeded generator."""
        v = _f32(vectors)
        n, dim = v.shape
        dsub = dim // num_subvectors
        rng = np.random.default_rng(seed)
        books = np.zeros((num_subvectors, 256, dsub), dtype=F)
        for s in range(num_subvectors):
            sub = v[:, s * dsub:(s + 1) * dsub]
            k = min(256, n)
            cents = [sub[rng.integers(n)].copy()]
            d2 = np.full(n, np.finfo(F).max, dtype=F)
            while len(cents) < k:
                d2 = np.minimum(d2, ((sub - cents[-1]) ** 2).sum(axis=1, dtype=F))
                tot = float(d2.sum(dtype=np.float64))
                if tot <= 0.0:
                    rest = np.flatnonzero(d2 > 0)
                    if rest.size == 0:
                        break
                    cents.append(sub[rng.choice(rest)].copy())
                    continue
                cents.append(sub[min(int(np.searchsorted(np.cumsum(d2, dtype=np.float64), rng.random() * tot)), n - 1)].copy())
            c = np.stack(cents).astype(F)
            assign = np.full(n, -1)
            for _ in range(iterations):
                dist = ((sub[:, None, :] - c[None, :, :]) ** 2).sum(axis=2, dtype=F)
                na = dist.argmin(axis=1)
                if np.array_equal(na, assign):
                    break
                assign = na
                nc = np.zeros_like(c)
                np.add.at(nc, assign, sub)
                cnt = np.bincount(assign, minlength=c.shape[0])
                for j in range(c.shape[0]):
                    nc[j] = nc[j] / F(cnt[j]) if cnt[j] else sub[rng.integers(n)]
                c = nc
            books[s, :c.shape[0]] = c
            if c.shape[0] < 256:  # fewer training rows than centroids: unused codes repeat the last centroid
                books[s, c.shape[0]:] = c[-1]
        return cls(books)

    @classmethod
    def train_device(cls, ctx: Context, vectors, num_subvectors: int, seed: int = 1, iterations: int = 20) -> "ProductQuantizer":
        """The same training on the GPU (needle_hip_pq_train): seeded splitmix64 draws, the reference's arithmetic."""
        v = _f32(vectors)
        n, dim = v.shape
        books = np.zeros((num_subvectors, 256, dim // num_subvectors), dtype=F)
        ncent = np.zeros(num_subvectors, dtype=np.uint32)
        _lib.check(_lib.lib().needle_hip_pq_train(ctx._h, v.ctypes.data, n, dim, num_subvectors, iterations, seed,
                                                  books.ctypes.data, ncent.ctypes.data))
        pq = cls(books)
        pq.centroids_per_subspace = ncent
        for s in range(num_subvectors):  # unused codes repeat the last centroid (never the first minimum)
            if 0 < ncent[s] < 256:
                pq.codebooks[s, ncent[s]:] = pq.codebooks[s, ncent[s] - 1]
        return pq

    def encode_device(self, ctx: Context, vectors) -> np.ndarray:
        """encode() on the GPU (needle_hip_pq_encode)."""
        v = _f32(vectors)
        out = np.zeros((v.shape[0], self.num_subvectors), dtype=np.uint8)
        _lib.check(_lib.lib().needle_hip_pq_encode(ctx._h, self.codebooks.ctypes.data, self.dimensions, self.num_subvectors,
                                                   None, v.ctypes.data, v.shape[0], out.ctypes.data))
        return out

    def encode(self, vectors) -> np.ndarray:
        """quantization.rs:390-446: per subspace the FIRST centroid at minimum squared distance."""
        v = _f32(vectors)
        single = v.ndim == 1
        v2 = v.reshape(1, -1) if single else v
        out = np.empty((v2.shape[0], self.num_subvectors), dtype=np.uint8)
        ds = self.subvector_dim
        for s in range(self.num_subvectors):
            sub = v2[:, s * ds:(s + 1) * ds]
            d = np.zeros((v2.shape[0], 256), dtype=F)
            for j in range(ds):  # dimension-ordered f32 sum, as the reference's iterator sum
                diff = (sub[:, j:j + 1] - self.codebooks[s, :, j][None, :]).astype(F)
                d = (d + diff * diff).astype(F)
            out[:, s] = d.argmin(axis=1)
        return out[0] if single else out

    def decode(self, codes) -> np.ndarray:  # quantization.rs:414-422
        c = np.asarray(codes, dtype=np.uint8)
        return np.concatenate([self.codebooks[s, c[s]] for s in range(self.num_subvectors)]).astype(F)


class BinaryQuantizer:
    """BinaryQuantizer, src/indexing/quantization.rs:536-594: one bit per dimension, set where the value
    exceeds the dimension's threshold (the training mean)."""

    def __init__(self, thresholds):
        self.thresholds = np.ascontiguousarray(thresholds, dtype=F)
        self.dimensions = self.thresholds.size

    @classmethod
    def train(cls, vectors) -> "BinaryQuantizer":  # quantization.rs:545-571: sequential f32 sums, one divide
        v = np.ascontiguousarray(vectors, dtype=F)
        if v.size == 0:
            return cls(np.zeros(0, dtype=F))
        acc = np.zeros(v.shape[1], dtype=F)
        for row in v:
            acc = (acc + row).astype(F)
        return cls(acc / F(v.shape[0]))

    def quantize(self, vectors) -> np.ndarray:  # quantization.rs:574-585: bit i % 8 of byte i / 8
        v = np.ascontiguousarray(vectors, dtype=F)
        single = v.ndim == 1
        v2 = v.reshape(1, -1) if single else v
        codes = np.packbits(v2 > self.thresholds[None, :], axis=1, bitorder="little")
        return codes[0] if single else codes

    def quantize_device(self, ctx: Context, vectors) -> np.ndarray:
        """The same codes computed on the GPU (needle_hip_binary_quantize)."""
        v = np.ascontiguousarray(vectors, dtype=F)
        out = np.zeros((v.shape[0], (self.dimensions + 7) // 8), dtype=np.uint8)
        _lib.check(_lib.lib().needle_hip_binary_quantize(ctx._h, self.thresholds.ctypes.data, v.ctypes.data, v.shape[0],
                                                         self.dimensions, out.ctypes.data))
        return out

    @staticmethod
    def hamming_distance(a, b) -> int:  # quantization.rs:588-593
        x = np.bitwise_xor(np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint8))
        return int(np.unpackbits(x).sum())


class _CodeIndex:
    def __init__(self):
        self._h = C.c_void_p()

    def close(self) -> None:
        _destroy_handle(self)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(_lib.lib().needle_hip_index_len(self._h))

    def search(self, queries, k: int, *, precision: str = "fp32", filter_mask=None):
        """(ids[q,k] uint64, dists[q,k] float32, counts[q] uint32), ascending by (distance, id)."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq = q.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), np.iinfo(np.uint64).max, dtype=np.uint64)
        dists = np.full((nq, kk), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fm = _mask_words(filter_mask, self.n)
        sp = _lib.SearchParams(k=k, nprobe=0, precision=PRECISION[precision], reserved=0,
                               filter_bitmask=fm.ctypes.data if fm is not None else None)
        _lib.check(_lib.lib().needle_hip_batch_search(self._h, q.ctypes.data if nq else None, nq, q.shape[1],
                                                      C.byref(sp), ids.ctypes.data, dists.ctypes.data,
                                                      counts.ctypes.data))
        return ids[:, :k], dists[:, :k], counts


class PqIndex(_CodeIndex):
    """Resident PQ code matrix scanned with the reference's distance-table arithmetic."""

    def __init__(self, ctx: Context, pq: ProductQuantizer, codes, valid=None):
        super().__init__()
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != pq.num_subvectors:
            raise ValueError("codes must be [n, num_subvectors]")
        self.ctx, self.pq, self.n = ctx, pq, c.shape[0]
        vm = _mask_words(valid, self.n)
        cb = np.ascontiguousarray(pq.codebooks, dtype=np.float32)
        _lib.check(_lib.lib().needle_hip_pq_build(ctx._h, c.ctypes.data if self.n else None, self.n, pq.dimensions,
                                                  pq.num_subvectors, cb.ctypes.data,
                                                  vm.ctypes.data if vm is not None else None, C.byref(self._h)))


class Sq8Index(_CodeIndex):
    """Resident u8 code matrix scanned with ScalarQuantizer::asymmetric_distance."""

    def __init__(self, ctx: Context, sq: ScalarQuantizer, codes, valid=None):
        super().__init__()
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != sq.dimensions:
            raise ValueError("codes must be [n, dimensions]")
        self.ctx, self.sq, self.n = ctx, sq, c.shape[0]
        vm = _mask_words(valid, self.n)
        _lib.check(_lib.lib().needle_hip_sq8_build(ctx._h, c.ctypes.data if self.n else None, self.n, sq.dimensions,
                                                   sq.min_vals.ctypes.data, sq.scale.ctypes.data,
                                                   vm.ctypes.data if vm is not None else None, C.byref(self._h)))


class BinaryIndex(_CodeIndex):
    """Resident binary code matrix scanned by Hamming distance; queries are f32 vectors, quantized on the
    device with the index's thresholds (src/indexing/compression_advisor.rs:658-671)."""

    def __init__(self, ctx: Context, bq: BinaryQuantizer, codes, valid=None):
        super().__init__()
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != (bq.dimensions + 7) // 8:
            raise ValueError("codes must be [n, ceil(dimensions / 8)]")
        self.ctx, self.bq, self.n = ctx, bq, c.shape[0]
        vm = _mask_words(valid, self.n)
        _lib.check(_lib.lib().needle_hip_binary_build(ctx._h, c.ctypes.data if self.n else None, self.n, bq.dimensions,
                                                      bq.thresholds.ctypes.data,
                                                      vm.ctypes.data if vm is not None else None, C.byref(self._h)))
