"""The arithmetic of a reference build with AVX2 enabled (src/distance.rs:481-591): the oracle's restatement against a
fold computed here by hand -- eight lane sums of exactly rounded fused multiply-adds (rational arithmetic, one rounding
to f32), hsum_avx's order, the tail after the fold -- and against the scalar order, from which it must differ."""
from fractions import Fraction

import numpy as np
import pytest

F = np.float32


def _round_f32(x: Fraction) -> np.float32:
    """x rounded once to the nearest f32, ties to even (finite, well inside the normal range)."""
    if x == 0:
        return F(0.0)
    c = F(float(x))  # within one ulp of the answer
    cands = [np.nextafter(c, F(-np.inf)), c, np.nextafter(c, F(np.inf))]
    return min(cands, key=lambda v: (abs(Fraction(float(v)) - x), int(np.array(v).view(np.uint32)) & 1))


def _fma(a, b, c) -> np.float32:
    return _round_f32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def _hsum(l):
    if len(l) == 4:                                  # vaddvq_f32: FADDP twice
        return F(F(l[0] + l[1]) + F(l[2] + l[3]))
    if len(l) == 16:                                 # hsum_avx512 (:619-627): low and high 256-bit halves first
        l = [F(l[j] + l[j + 8]) for j in range(8)]
    s = [F(l[j] + l[j + 4]) for j in range(4)]      # _mm_add_ps(low half, high half)
    t0, t1 = F(s[0] + s[2]), F(s[1] + s[3])          # + movehl
    return F(t0 + t1)                                # + shuffle(1)


def _by_hand(kind, a, b, width=8):
    n = a.size
    lanes = [F(0.0)] * width
    for c in range(n // width):
        for j in range(width):
            x, y = a[c * width + j], b[c * width + j]
            if kind == "dot":
                lanes[j] = _fma(x, y, lanes[j])
            elif kind == "l2":
                d = F(x - y)
                lanes[j] = _fma(d, d, lanes[j])
            else:
                lanes[j] = F(lanes[j] + np.abs(F(x - y)))
    r = _hsum(lanes)
    for i in range(n // width * width, n):
        x, y = a[i], b[i]
        if kind == "dot":
            r = F(r + F(x * y))
        elif kind == "l2":
            d = F(x - y)
            r = F(r + F(d * d))
        else:
            r = F(r + np.abs(F(x - y)))
    return r


def _bits(x):
    return int(np.array(x, dtype=F).view(np.uint32))


@pytest.mark.parametrize("dim", [1, 7, 8, 9, 15, 16, 17, 40, 100, 384])
def test_oracle_follows_the_hand_computed_eight_lane_fold(oracle, dim):
    rng = np.random.default_rng(dim)
    for _ in range(3):
        a, b = rng.standard_normal(dim).astype(F), rng.standard_normal(dim).astype(F)
        assert _bits(oracle.dot_avx2(a, b)) == _bits(_by_hand("dot", a, b))
        assert _bits(oracle.euclidean_squared_avx2(a, b)) == _bits(_by_hand("l2", a, b))
        assert _bits(oracle.manhattan_avx2(a, b)) == _bits(_by_hand("l1", a, b))
        assert _bits(oracle.dot_avx512(a, b)) == _bits(_by_hand("dot", a, b, 16))
        assert _bits(oracle.euclidean_squared_avx512(a, b)) == _bits(_by_hand("l2", a, b, 16))
        assert _bits(oracle.manhattan_avx512(a, b)) == _bits(_by_hand("l1", a, b, 16))
        assert _bits(oracle.dot_neon(a, b)) == _bits(_by_hand("dot", a, b, 4))
        assert _bits(oracle.euclidean_squared_neon(a, b)) == _bits(_by_hand("l2", a, b, 4))
        assert _bits(oracle.manhattan_neon(a, b)) == _bits(_by_hand("l1", a, b, 4))


def test_fused_rounding_is_what_decides_a_lane():
    """A product whose exact value a separate multiply would round away: 1 + 2^-12 squared is 1 + 2^-11 + 2^-24; minus
    (1 + 2^-11) the fused form keeps 2^-24, the two-rounding form returns 0."""
    x = F(1.0 + 2.0 ** -12)
    c = F(-(1.0 + 2.0 ** -11))
    assert float(_fma(x, x, c)) == 2.0 ** -24
    assert float(F(F(x * x) + c)) == 0.0


def test_the_order_differs_from_the_scalar_build_on_a_gaussian_pair(oracle):
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal(384).astype(F), rng.standard_normal(384).astype(F)
    assert _bits(oracle.dot(a, b)) != _bits(oracle.dot_avx2(a, b))
    assert _bits(oracle.euclidean_squared(a, b)) != _bits(oracle.euclidean_squared_avx2(a, b))
    assert abs(oracle.dot(a, b) - oracle.dot_avx2(a, b)) < 1e-4
    av = oracle.AVX2_ORDER
    differing = sum(_bits(oracle.compute(m, a, b)) != _bits(oracle.compute(m | av, a, b)) for m in range(5))
    assert differing >= 3  # cosine, euclidean, dot, ... on this pair
    for m in (5, 6):       # hamming, chebyshev: no SIMD form
        assert _bits(oracle.compute(m, a, b)) == _bits(oracle.compute(m | av, a, b))
        assert _bits(oracle.compute(m, a, b)) == _bits(oracle.compute(m | oracle.AVX512_ORDER, a, b))
    # the three builds are three different arithmetics (on some pair: the sums differ in their last bits only)
    rng = np.random.default_rng(4)
    pairs = [(rng.standard_normal(384).astype(F), rng.standard_normal(384).astype(F)) for _ in range(20)]
    assert any(_bits(oracle.dot_avx2(x, y)) != _bits(oracle.dot_avx512(x, y)) for x, y in pairs)
    assert any(_bits(oracle.dot(x, y)) != _bits(oracle.dot_avx512(x, y)) for x, y in pairs)
    assert any(_bits(oracle.dot_neon(x, y)) != _bits(oracle.dot_avx2(x, y)) for x, y in pairs)
    x, y = pairs[0]
    assert _bits(oracle.compute(2 | oracle.NEON_ORDER, x, y)) == _bits(np.sqrt(F(oracle.euclidean_squared_neon(x, y))))
    assert _bits(oracle.compute(5 | oracle.NEON_ORDER, x, y)) == _bits(oracle.compute(5, x, y))


def test_compute_composes_the_metric_from_the_lane_sums(oracle):
    rng = np.random.default_rng(2)
    av = oracle.AVX2_ORDER
    for dim in (5, 64, 99):
        a, b = rng.standard_normal(dim).astype(F), rng.standard_normal(dim).astype(F)
        dot, qq, bb = F(oracle.dot_avx2(a, b)), F(oracle.dot_avx2(a, a)), F(oracle.dot_avx2(b, b))
        cos = F(F(1.0) - F(dot / F(np.sqrt(qq) * np.sqrt(bb))))
        assert _bits(oracle.compute(0 | av, a, b)) == _bits(cos)
        assert _bits(oracle.compute(1 | av, a, b)) == _bits(F(F(1.0) - dot))
        assert _bits(oracle.compute(2 | av, a, b)) == _bits(np.sqrt(F(oracle.euclidean_squared_avx2(a, b))))
        assert _bits(oracle.compute(3 | av, a, b)) == _bits(-dot)
        assert _bits(oracle.compute(4 | av, a, b)) == _bits(oracle.manhattan_avx2(a, b))
        wide = oracle.AVX512_ORDER
        dot, qq, bb = F(oracle.dot_avx512(a, b)), F(oracle.dot_avx512(a, a)), F(oracle.dot_avx512(b, b))
        assert _bits(oracle.compute(0 | wide, a, b)) == _bits(F(F(1.0) - F(dot / F(np.sqrt(qq) * np.sqrt(bb)))))
        assert _bits(oracle.compute(2 | wide, a, b)) == _bits(np.sqrt(F(oracle.euclidean_squared_avx512(a, b))))
        assert _bits(oracle.compute(4 | wide, a, b)) == _bits(oracle.manhattan_avx512(a, b))
    z = np.zeros(16, dtype=F)
    assert oracle.compute(0 | av, z, rng.standard_normal(16).astype(F)) == 1.0  # zero norm (distance.rs:159-161)


def test_flat_search_and_normalize_in_that_order(oracle):
    rng = np.random.default_rng(3)
    base = rng.standard_normal((300, 40)).astype(F)
    queries = rng.standard_normal((5, 40)).astype(F)
    av = oracle.AVX2_ORDER
    ids, dists, counts = oracle.flat_batch_search(queries, base, 7, 2 | av)
    for q in range(5):
        d = np.array([oracle.compute(2 | av, queries[q], r) for r in base], dtype=F)
        order = np.lexsort((np.arange(300), d))[:7]
        assert np.array_equal(ids[q], order.astype(np.uint64))
        assert np.array_equal(dists[q].view(np.uint32), d[order].view(np.uint32))
    v = rng.standard_normal(33).astype(F)
    inv = F(F(1.0) / np.sqrt(F(oracle.dot_avx2(v, v))))
    assert np.array_equal(oracle.normalized(v, avx2=True).view(np.uint32), (v * inv).astype(F).view(np.uint32))
