"""Host-side mirror of the reference's half-precision storage types (src/indexing/float16.rs): the `F16` and
`Bf16` conversions over numpy arrays of raw 16-bit patterns, the vector operations of `F16Vector` / `Bf16Vector`,
and `HalfFlatIndex`, a flat index built from half rows that are widened on the GPU (needle_hip_flat_build_half).

`F16::from_f32` TRUNCATES the mantissa (float16.rs:61-98) -- it is not IEEE round-to-nearest, so numpy's
`astype(float16)` is not a substitute; `Bf16::from_f32` truncates, `from_f32_round` rounds to nearest even.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .distance import DistanceFunction
from .index import Context, FlatIndex, _mask_words

F = np.float32
HALF_F16, HALF_BF16 = 0, 1


def f16_from_f32(values) -> np.ndarray:  # F16::from_f32, float16.rs:61-98
    bits = np.ascontiguousarray(values, dtype=F).view(np.uint32).astype(np.int64)
    sign = (bits >> 31) & 1
    exp = (bits >> 23) & 0xFF
    mant = bits & 0x7FFFFF
    new_exp = exp - 127 + 15
    shift = np.clip(1 - new_exp + 13, 0, 63)
    sub = np.where(new_exp < -10, 0, (mant | 0x800000) >> shift)
    out = np.where(new_exp <= 0, sub, np.where(new_exp >= 31, 0x7C00, (new_exp << 10) | (mant >> 13)))
    out = np.where(exp == 255, np.where(mant == 0, 0x7C00, 0x7E00), out)
    return ((sign << 15) | out).astype(np.uint16)


def f16_to_f32(bits) -> np.ndarray:  # F16::to_f32, float16.rs:101-136
    b = np.ascontiguousarray(bits, dtype=np.uint16).astype(np.int64)
    sign, exp, mant = (b >> 15) & 1, (b >> 10) & 0x1F, b & 0x3FF
    # subnormal: shift the leading one up to bit 10, the exponent drops by the shifts made
    lead = np.zeros_like(mant)
    m = mant.copy()
    for _ in range(10):
        grow = (m != 0) & ((m & 0x400) == 0)
        m = np.where(grow, m << 1, m)
        lead = lead + grow
    sub = np.where(mant == 0, 0, ((127 - 15 + 1 - lead) << 23) | ((m & 0x3FF) << 13))
    norm = ((exp - 15 + 127) << 23) | (mant << 13)
    special = np.where(mant == 0, 0x7F800000, 0x7FC00000)
    out = np.where(exp == 0, sub, np.where(exp == 31, special, norm))
    return ((sign << 31) | out).astype(np.uint32).view(F)


def bf16_from_f32(values) -> np.ndarray:  # Bf16::from_f32, float16.rs:202-206: the upper 16 bits
    return (np.ascontiguousarray(values, dtype=F).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_f32_round(values) -> np.ndarray:  # Bf16::from_f32_round, float16.rs:209-220
    bits = np.ascontiguousarray(values, dtype=F).view(np.uint32).astype(np.int64)
    upper = bits >> 16
    up = (((bits >> 15) & 1) != 0) & (((bits & 0x7FFF) != 0) | ((upper & 1) != 0))
    return ((upper + up) & 0xFFFF).astype(np.uint16)  # wrapping_add


def bf16_to_f32(bits) -> np.ndarray:  # Bf16::to_f32, float16.rs:223-225
    return (np.ascontiguousarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(F)


def _seq_sum(terms: np.ndarray) -> np.float32:  # Iterator::sum::<f32>(): sequential, from -0.0
    s = F(-0.0)
    for t in terms:
        s = F(s + t)
    return s


class _HalfVector:
    KIND = HALF_F16

    def __init__(self, bits):
        self.data = np.ascontiguousarray(bits, dtype=np.uint16)

    def to_f32(self) -> np.ndarray:
        return f16_to_f32(self.data) if self.KIND == HALF_F16 else bf16_to_f32(self.data)

    def __len__(self) -> int:
        return int(self.data.size)

    def is_empty(self) -> bool:
        return self.data.size == 0

    def memory_bytes(self) -> int:  # float16.rs:328-330, 408-410
        return int(self.data.size) * 2

    def euclidean_distance(self, other) -> float:  # float16.rs:294-304, 375-384
        d = (self.to_f32() - other.to_f32()).astype(F)
        return float(np.sqrt(_seq_sum((d * d).astype(F)), dtype=F))

    def dot(self, other) -> float:  # float16.rs:307-312, 387-392
        return float(_seq_sum((self.to_f32() * other.to_f32()).astype(F)))

    def cosine_similarity(self, other) -> float:  # float16.rs:315-326, 395-406: a similarity; zero norm -> 0.0
        a, b = self.to_f32(), other.to_f32()
        na = np.sqrt(_seq_sum((a * a).astype(F)), dtype=F)
        nb = np.sqrt(_seq_sum((b * b).astype(F)), dtype=F)
        if na == 0.0 or nb == 0.0:
            return 0.0
        return float(F(self.dot(other)) / F(na * nb))


class F16Vector(_HalfVector):  # float16.rs:261-331
    KIND = HALF_F16

    @classmethod
    def from_f32(cls, values) -> "F16Vector":
        return cls(f16_from_f32(values))


class Bf16Vector(_HalfVector):  # float16.rs:335-411
    KIND = HALF_BF16

    @classmethod
    def from_f32(cls, values) -> "Bf16Vector":
        return cls(bf16_from_f32(values))

    @classmethod
    def from_f32_round(cls, values) -> "Bf16Vector":
        return cls(bf16_from_f32_round(values))


def memory_savings(dimensions: int, num_vectors: int):  # float16.rs:461-470 (checked multiplication -> None on overflow)
    f32_size = dimensions * num_vectors * 4
    f16_size = dimensions * num_vectors * 2
    if f32_size >= 1 << 64:
        return None
    return f32_size, f16_size, (float(F(1.0) - F(f16_size) / F(f32_size)) * 100.0 if f32_size else 0.0)


class HalfFlatIndex(FlatIndex):
    """A flat index whose rows arrive in half precision and are widened on the device."""

    def __init__(self, ctx: Context, bits, kind: int, metric: DistanceFunction, valid=None):
        b = np.ascontiguousarray(bits, dtype=np.uint16)
        if b.ndim != 2:
            raise ValueError("rows must be [n, dim] raw 16-bit patterns")
        self.ctx, self.metric, self.n, self.dim = ctx, DistanceFunction(metric), b.shape[0], b.shape[1]
        self._h = C.c_void_p()
        vm = _mask_words(valid, self.n)
        _lib.check(_lib.lib().needle_hip_flat_build_half(ctx._h, b.ctypes.data if self.n else None, self.n, self.dim, kind,
                                                         int(self.metric), _lib.ARITH_DISTANCE_RS,
                                                         vm.ctypes.data if vm is not None else None, C.byref(self._h)))
