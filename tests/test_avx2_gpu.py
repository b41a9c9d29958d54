"""NEEDLE_ARITH_AVX2 / NEEDLE_ARITH_AVX512 on the device: flat search, the distance operators and normalize report the
bits a reference build with AVX2 / AVX-512 enabled computes (src/distance.rs:481-738) -- checked against the oracle's
restatement of those orders, and seen to differ from the scalar build's bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint32)


def _check(oracle, got, base, queries, k, code, valid=None):
    oi, od, oc = oracle.flat_batch_search(queries, base, k, code, valid=valid)
    ids, dists, counts = got
    assert np.array_equal(counts, oc)
    for i in range(queries.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(_bits(dists[i, :c]), _bits(od[i, :c])), i


def _build(oracle, name):
    from needle_amd import _lib

    return {"avx2": (_lib.ARITH_AVX2, oracle.AVX2_ORDER), "avx512": (_lib.ARITH_AVX512, oracle.AVX512_ORDER),
            "neon": (_lib.ARITH_NEON, oracle.NEON_ORDER)}[name]


@pytest.mark.parametrize("build", ["avx2", "avx512", "neon"])
@pytest.mark.parametrize("metric", ["Cosine", "CosineNormalized", "Euclidean", "DotProduct", "Manhattan", "Hamming", "Chebyshev"])
@pytest.mark.parametrize("dim", [7, 100, 384])
def test_flat_search_reports_the_simd_builds_bits(ctx, oracle, metric, dim, build):
    import needle_amd as na

    arith, order = _build(oracle, build)

    rng = np.random.default_rng(dim * 7 + len(metric))
    n, nq, k = 3000, 60, 10
    base = rng.standard_normal((n, dim)).astype(F)
    queries = rng.standard_normal((nq, dim)).astype(F)
    if metric == "CosineNormalized":
        base, queries = oracle.normalize_rows(base), oracle.normalize_rows(queries)
    if metric == "Hamming":
        base[rng.random(base.shape) < 0.5] = 0.0
        queries[rng.random(queries.shape) < 0.5] = 0.0
    base[1500] = base[3]          # ties: the lower row wins
    queries[0] = base[3]
    base[77] = 0.0                # a zero norm
    valid = rng.random(n) > 0.1
    m = na.DistanceFunction[metric]
    code = int(m) | order
    ix = na.FlatIndex(ctx, base, m, arith=arith, valid=valid)
    for prec in ("bf16", "fp32"):
        _check(oracle, ix.search(queries, k, precision=prec), base, queries, k, code, valid=valid)
    _check(oracle, ix.search(queries[:8], 300), base, queries[:8], 300, code, valid=valid)  # beyond the candidate budget: the exact scan
    if metric in ("Cosine", "Euclidean", "DotProduct", "Manhattan") and dim >= 100:
        # the mode is live: the scalar build's bits differ somewhere among the reported distances
        plain = na.FlatIndex(ctx, base, m, valid=valid)
        a, b = ix.search(queries, k)[1], plain.search(queries, k)[1]
        assert not np.array_equal(_bits(a), _bits(b))
        plain.close()
    ix.close()


def test_append_keeps_the_arithmetic(ctx, oracle):
    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(5)
    base = rng.standard_normal((2000, 96)).astype(F)
    queries = rng.standard_normal((40, 96)).astype(F)
    ix = na.FlatIndex(ctx, base[:1200], na.DistanceFunction.Euclidean, arith=_lib.ARITH_AVX2)
    ix.append(base[1200:])
    _check(oracle, ix.search(queries, 5, precision="bf16"), base, queries, 5, 2 | oracle.AVX2_ORDER)
    ix.close()


@pytest.mark.parametrize("build", ["avx2", "avx512", "neon"])
@pytest.mark.parametrize("dim", [5, 8, 33, 384])
def test_operators_in_the_simd_build_orders(ctx, oracle, dim, build):
    import needle_amd as na

    arith, av = _build(oracle, build)

    rng = np.random.default_rng(dim)
    q = rng.standard_normal((19, dim)).astype(F)
    b = rng.standard_normal((150, dim)).astype(F)
    b[4] = 0.0
    for metric in (na.DistanceFunction.Cosine, na.DistanceFunction.Euclidean, na.DistanceFunction.DotProduct,
                   na.DistanceFunction.Manhattan, na.DistanceFunction.Chebyshev):
        got = ctx.distance_matrix(metric, q, b, arith)
        want = np.array([[oracle.compute(int(metric) | av, qi, bi) for bi in b] for qi in q], dtype=F)
        assert np.array_equal(_bits(got), _bits(want)), metric
    for metric in (na.DistanceFunction.Cosine, na.DistanceFunction.Euclidean, na.DistanceFunction.DotProduct):
        got = ctx.batch_distance(metric, q[0], b, arith)
        want = np.array([oracle.compute(int(metric) | av, q[0], bi) for bi in b], dtype=F)
        if metric == na.DistanceFunction.DotProduct:
            want = -want  # batch_dot_product reports the product itself
        assert np.array_equal(_bits(got), _bits(want)), metric
    idx, val, cnt = ctx.multi_query_search(na.DistanceFunction.Euclidean, q, b, 6, arith)
    oi, od, oc = oracle.flat_batch_search(q, b, 6, 2 | av)
    assert np.array_equal(np.asarray(idx).astype(np.uint64), oi) and np.array_equal(_bits(val), _bits(od))
    got = ctx.normalize(b, arith)
    want = np.stack([oracle.normalized(r, avx2=build == "avx2", avx512=build == "avx512", neon=build == "neon") for r in b])
    assert np.array_equal(_bits(got), _bits(want))
