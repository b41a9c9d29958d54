"""Half-precision storage (src/indexing/float16.rs): the oracle's conversions against the reference's own tests,
the numpy mirror against the oracle over EVERY 16-bit pattern, and -- on the GPU -- the device widening checked
exhaustively plus a flat search over half rows against the oracle's search over the widened rows."""
import numpy as np
import pytest

F = np.float32
ALL16 = np.arange(65536, dtype=np.uint32).astype(np.uint16)


def _same_f32(a, b):  # bit equality, every NaN counted as one value (the reference canonicalises f16 NaNs)
    a, b = np.asarray(a, dtype=F), np.asarray(b, dtype=F)
    return bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)]))


def test_reference_kats(oracle):  # float16.rs:472-600
    o = oracle
    vals = np.array([0.0, 1.0, -1.0, 0.5, 100.0, -0.001, 65504.0], dtype=F)           # test_f16_conversion
    back = o.half_to_f32(o.HALF_F16, o.half_from_f32(o.HALF_F16, vals))
    assert np.all(np.abs(vals - back) < np.abs(vals) * 0.01 + 0.001)
    sp = o.half_from_f32(o.HALF_F16, np.array([0.0, np.inf, -np.inf, np.nan], dtype=F))  # test_f16_special_values
    assert sp[0] & 0x7FFF == 0 and sp[1] == 0x7C00 and sp[2] == 0xFC00 and (sp[3] & 0x7C00) == 0x7C00 and (sp[3] & 0x3FF) != 0
    vals = np.array([0.0, 1.0, -1.0, 0.5, 100.0, -0.001, 1e10, 1e-10], dtype=F)        # test_bf16_conversion
    back = o.half_to_f32(o.HALF_BF16, o.half_from_f32(o.HALF_BF16, vals))
    assert np.all(np.abs((vals - back)[vals != 0] / vals[vals != 0]) < 0.01) and back[0] == 0.0
    sp = o.half_from_f32(o.HALF_BF16, np.array([0.0, np.inf, np.nan], dtype=F))          # test_bf16_special_values
    assert sp[0] == 0 and sp[1] == 0x7F80 and (sp[2] & 0x7F80) == 0x7F80 and (sp[2] & 0x7F) != 0
    assert o.half_to_f32(o.HALF_BF16, np.array([0x3F80], dtype=np.uint16))[0] == 1.0      # Bf16::ONE, :181
    # truncation, not rounding (float16.rs:96-97): 1 + 2^-10 + 2^-11 keeps only the 2^-10 step
    assert o.half_from_f32(o.HALF_F16, np.array([1.0 + 2.0 ** -10 + 2.0 ** -11], dtype=F))[0] == 0x3C01
    # from_f32_round (float16.rs:209-220), round to nearest even on the 16 dropped bits:
    #   1 + 2^-8 + 2^-9   above halfway                      -> up to 0x3F81
    #   1 + 2^-8          exactly halfway, upper 0x3F80 even -> stays
    #   1 + 2^-7 + 2^-8   exactly halfway, upper 0x3F81 odd  -> up to 0x3F82
    rne = np.array([1.0 + 2.0 ** -8 + 2.0 ** -9, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7 + 2.0 ** -8], dtype=F)
    assert o.half_from_f32(o.HALF_BF16, rne, rounded=True).tolist() == [0x3F81, 0x3F80, 0x3F82]
    assert o.half_from_f32(o.HALF_BF16, rne).tolist() == [0x3F80, 0x3F80, 0x3F81]  # from_f32 only truncates


def test_numpy_mirror_equals_oracle_on_every_pattern(oracle):
    from needle_amd import float16 as h

    assert _same_f32(h.f16_to_f32(ALL16), oracle.half_to_f32(oracle.HALF_F16, ALL16))
    assert _same_f32(h.bf16_to_f32(ALL16), oracle.half_to_f32(oracle.HALF_BF16, ALL16))
    rng = np.random.default_rng(3)
    vals = np.concatenate([
        rng.integers(0, 2 ** 32, 300_000, dtype=np.uint64).astype(np.uint32).view(F),     # any bit pattern, NaNs included
        (rng.standard_normal(50_000) * 10.0 ** rng.uniform(-9, 6, 50_000)).astype(F),    # f16 subnormal .. overflow range
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65520.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -26, 6e-8], dtype=F)])
    assert np.array_equal(h.f16_from_f32(vals), oracle.half_from_f32(oracle.HALF_F16, vals))
    assert np.array_equal(h.bf16_from_f32(vals), oracle.half_from_f32(oracle.HALF_BF16, vals))
    assert np.array_equal(h.bf16_from_f32_round(vals), oracle.half_from_f32(oracle.HALF_BF16, vals, rounded=True))
    # round trip: widening then converting back is the identity on every non-NaN pattern
    f16_ok = ~np.isnan(h.f16_to_f32(ALL16))
    assert np.array_equal(h.f16_from_f32(h.f16_to_f32(ALL16))[f16_ok], ALL16[f16_ok])
    assert np.array_equal(h.bf16_from_f32(h.bf16_to_f32(ALL16)), ALL16)


def test_half_vectors(oracle):  # float16.rs:536-600
    from needle_amd.float16 import Bf16Vector, F16Vector, memory_savings

    v = F16Vector.from_f32([1.0, 2.0, 3.0, 4.0])
    assert len(v) == 4 and v.memory_bytes() == 8 and np.all(np.abs(v.to_f32() - [1, 2, 3, 4]) < 0.01)
    assert abs(F16Vector.from_f32([1.0, 0.0, 0.0]).euclidean_distance(F16Vector.from_f32([0.0, 1.0, 0.0])) - 2 ** 0.5) < 0.1
    assert abs(Bf16Vector.from_f32([1.0, 2.0, 3.0]).dot(Bf16Vector.from_f32([4.0, 5.0, 6.0])) - 32.0) < 1.0
    a = F16Vector.from_f32([1.0, 0.0])
    assert abs(a.cosine_similarity(F16Vector.from_f32([1.0, 0.0])) - 1.0) < 0.01
    assert abs(a.cosine_similarity(F16Vector.from_f32([0.0, 1.0]))) < 0.01
    assert a.cosine_similarity(F16Vector.from_f32([0.0, 0.0])) == 0.0
    f32_size, f16_size, savings = memory_savings(384, 1_000_000)
    assert f32_size == 384 * 1_000_000 * 4 and f16_size == 384 * 1_000_000 * 2 and abs(savings - 50.0) < 0.1
    # the vector distance is the f32 arithmetic of src/distance.rs on the widened values
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(70).astype(F), rng.standard_normal(70).astype(F)
    hx, hy = F16Vector.from_f32(x), F16Vector.from_f32(y)
    assert F(hx.euclidean_distance(hy)) == F(oracle.compute(2, hx.to_f32(), hy.to_f32()))
    assert F(hx.dot(hy)) == F(-oracle.compute(3, hx.to_f32(), hy.to_f32()))


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_device_widening_is_exact_on_every_finite_pattern(ctx, oracle, kind):
    """A 1-d index over all finite patterns, queried at each pattern's correct f32 value with k = 1 under the
    Manhattan metric: |q - v| is exact, subnormals included (a squared difference of the smallest bf16 values would
    underflow to 0 and tie them all), so distance 0 singles out the row holding exactly that value -- for -0 the +0
    row: same value, lower id -- unless the device widened a pattern wrongly."""
    import needle_amd as na
    from needle_amd.float16 import HalfFlatIndex

    want = oracle.half_to_f32(kind, ALL16)
    finite = np.isfinite(want)
    bits = ALL16[finite]
    vals = want[finite]
    ix = HalfFlatIndex(ctx, bits.reshape(-1, 1), kind, na.DistanceFunction.Manhattan)
    ids, dists, counts = ix.search(vals.reshape(-1, 1), 1)
    first_with_value = {}
    for i, v in enumerate(vals):  # +0 and -0 are one value; the lower row id wins the tie
        first_with_value.setdefault(float(v), i)
    expect = np.array([first_with_value[float(v)] for v in vals], dtype=np.uint64)
    assert np.all(counts == 1) and np.array_equal(ids[:, 0], expect) and np.all(dists[:, 0] == 0.0)
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,metric", [(0, "Cosine"), (0, "Euclidean"), (1, "DotProduct"), (1, "Euclidean")])
def test_search_over_half_rows(ctx, oracle, kind, metric):
    import needle_amd as na
    from needle_amd import float16 as h
    from needle_amd.float16 import HalfFlatIndex

    rng = np.random.default_rng(7 + kind)
    rows = (rng.standard_normal((9000, 96)) * 10.0 ** rng.uniform(-6, 1, (9000, 1))).astype(F)  # into the f16 subnormals
    bits = h.f16_from_f32(rows) if kind == 0 else h.bf16_from_f32_round(rows)
    wide = oracle.half_to_f32(kind, bits)
    queries = rng.standard_normal((40, 96)).astype(F)
    m = na.DistanceFunction[metric]
    valid = rng.random(9000) > 0.2
    ix = HalfFlatIndex(ctx, bits, kind, m, valid=valid)
    for precision in ("fp32", "bf16"):
        ids, dists, counts = ix.search(queries, 10, precision=precision)
        oi, od, oc = oracle.flat_batch_search(queries, wide, 10, int(m), valid=valid)
        assert np.array_equal(counts, oc) and np.array_equal(ids, oi) and np.array_equal(dists.view(np.uint32), od.view(np.uint32))
    ix.close()
    nan_bits = bits.copy()
    nan_bits[5, 3] = 0x7E00 if kind == 0 else 0x7FC0
    with pytest.raises(na.InvalidVector):
        HalfFlatIndex(ctx, nan_bits, kind, m)
