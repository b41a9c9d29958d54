"""The bf16 first pass (precision="bf16") returns the SAME hits as the fp32 one: the reduced
precision only widens the candidate filter; reported ids and distances stay bit-equal to the
oracle's reference-order arithmetic, with the bound check (and the exact scan behind it) covering
every query the coarser scores cannot decide."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32
U64MAX = np.iinfo(np.uint64).max
METRICS4 = ["Cosine", "CosineNormalized", "Euclidean", "DotProduct"]


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _data(seed, n, d, q, unit=False):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, d)).astype(F)
    qq = rng.standard_normal((q, d)).astype(F)
    if unit:
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        qq /= np.linalg.norm(qq, axis=1, keepdims=True)
    return b.astype(F), qq.astype(F)


def _check(oracle, got, base, queries, k, metric, valid=None, filt=None):
    ids, dists, counts = got
    oi, od, oc = oracle.flat_batch_search(queries, base, k, int(metric), valid=valid, filt=filt)
    assert np.array_equal(counts, oc), (counts[:8], oc[:8])
    for i in range(queries.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (i, dists[i, :c], od[i, :c])
        assert np.all(ids[i, c:] == U64MAX) and np.all(np.isinf(dists[i, c:]))


# every register-resident query layout: dpad 128, 256, 384 (eight waves of two query tiles), 512, 768, 1024 (four waves
# of two), 1536 (four waves of one)
@pytest.mark.parametrize("metric", METRICS4)
@pytest.mark.parametrize("shape", [(3000, 100, 130, 10), (517, 33, 3, 1), (9000, 200, 300, 10), (6000, 384, 270, 10),
                                   (4000, 500, 140, 5), (3000, 768, 40, 10), (20000, 1024, 150, 10), (2500, 900, 70, 3),
                                   (20000, 1536, 100, 10), (3000, 1300, 33, 10)])
def test_bf16_shapes_all_metrics(ctx, oracle, metric, shape):
    import needle_amd as na

    n, d, q, k = shape
    base, queries = _data(31 + d, n, d, q, unit=(metric == "CosineNormalized"))
    m = na.DistanceFunction[metric]
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, k, precision="bf16"), base, queries, k, m)
    ix.close()


@pytest.mark.parametrize("metric", ["Cosine", "Euclidean"])
def test_bf16_long_slices_seeded_threshold_and_masks(ctx, oracle, metric):
    import needle_amd as na

    n, d, q, k = 200_000, 48, 300, 10
    base, queries = _data(41, n, d, q)
    m = na.DistanceFunction[metric]
    rng = np.random.default_rng(3)
    valid = rng.random(n) > 0.2
    filt = rng.random(n) < 0.1
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, k, precision="bf16"), base, queries, k, m)
    ix.set_valid_mask(valid)
    _check(oracle, ix.search(queries, k, precision="bf16", filter_mask=filt), base, queries, k, m, valid=valid, filt=filt)
    ix.close()


def test_bf16_readme_shape_scaled(ctx, oracle):
    import needle_amd as na

    base, queries = _data(2, 60000, 384, 600, unit=True)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    ctx.reset_metrics()
    _check(oracle, ix.search(queries, 10, precision="bf16"), base, queries, 10, na.DistanceFunction.Cosine)
    # well separated random embeddings: the bf16 filter decides nearly every query on its own
    m = ctx.metrics()
    assert m["refined_queries"] + m["fallback_queries"] <= 6
    ix.close()


def test_bf16_sorted_base_and_large_k(ctx, oracle):
    import needle_amd as na

    base, queries = _data(22, 60_000, 24, 8)
    order = np.argsort(-((base - queries[0]) ** 2).sum(axis=1))
    base = np.ascontiguousarray(base[order])  # farthest first: every tile beats the running threshold
    m = na.DistanceFunction.Euclidean
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, 10, precision="bf16"), base, queries, 10, m)
    _check(oracle, ix.search(queries, 150, precision="bf16"), base, queries, 150, m)  # kprime 208
    _check(oracle, ix.search(queries, 230, precision="bf16"), base, queries, 230, m)  # bf16 slack does not fit: fp32
    ix.close()


def test_bf16_close_rows_fall_back(ctx, oracle):
    """Rows closer together than bf16 resolves: the bound check must flag the query, which is then
    served by the fp32 first pass (and, where that cannot decide either, by the exact scan)."""
    import needle_amd as na

    rng = np.random.default_rng(7)
    d = 64
    centre = rng.standard_normal(d).astype(F)
    base = rng.standard_normal((5000, d)).astype(F)
    base[500:900] = centre + (rng.standard_normal((400, d)) * 1e-3).astype(F)
    queries = np.stack([centre, rng.standard_normal(d).astype(F)])
    for m in (na.DistanceFunction.Cosine, na.DistanceFunction.Euclidean):
        ctx.reset_metrics()
        ix = na.FlatIndex(ctx, base, m)
        _check(oracle, ix.search(queries, 10, precision="bf16"), base, queries, 10, m)
        ix.close()
        m2 = ctx.metrics()
        assert m2["refined_queries"] + m2["fallback_queries"] >= 1


def test_bf16_ties_masks_and_dimension_beyond_registers(ctx, oracle):
    import needle_amd as na

    base, queries = _data(5, 2000, 32, 4)
    base[100] = base[7]
    base[1500] = base[7]
    queries[0] = base[7]
    m = na.DistanceFunction.Euclidean
    ix = na.FlatIndex(ctx, base, m)
    got = ix.search(queries, 6, precision="bf16")
    _check(oracle, got, base, queries, 6, m)
    assert got[0][0, :3].tolist() == [7, 100, 1500]
    none = np.zeros(2000, dtype=bool)
    assert ix.search(queries, 5, precision="bf16", filter_mask=none)[2].tolist() == [0] * 4
    ix.close()
    base, queries = _data(6, 1500, 1000, 9)  # wider than the bf16 kernel holds: served by the fp32 scan
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    _check(oracle, ix.search(queries, 5, precision="bf16"), base, queries, 5, na.DistanceFunction.Cosine)
    ix.close()


def test_small_base_selection_corner_cases(ctx, oracle):
    """Bases small enough for the dense score matrix: the selection narrows every row of scores to the rows above a
    pivot.  Here the narrowing cannot work: thousands of rows tie at the top (more survivors than its buffer holds),
    and masks that leave live rows in a few lanes' columns only (no pivot); both must still return the oracle's rows."""
    import needle_amd as na

    rng = np.random.default_rng(21)
    n, dim = 6000, 40
    base = rng.standard_normal((n, dim)).astype(F)
    base[500:4500] = base[7]                      # 4000 identical rows: ties broken by id
    queries = rng.standard_normal((70, dim)).astype(F)
    queries[0] = base[7]
    queries[1] = -base[7]                          # the tied block at the far end
    for metric in (na.DistanceFunction.Euclidean, na.DistanceFunction.Cosine, na.DistanceFunction.DotProduct):
        ix = na.FlatIndex(ctx, base, metric)
        for k in (1, 10, 100):
            _check(oracle, ix.search(queries, k, precision="bf16"), base, queries, k, metric)
        # live rows only where row % 256 < 8: two of the selection's 64 lanes see anything
        keep = (np.arange(n) % 256) < 8
        for k in (10, 150):
            _check(oracle, ix.search(queries, k, precision="bf16", filter_mask=keep), base, queries, k, metric, filt=keep)
        few = np.zeros(n, dtype=bool)
        few[[3, 4000, 5999]] = True               # fewer live rows than k
        _check(oracle, ix.search(queries, 10, precision="bf16", filter_mask=few), base, queries, 10, metric, filt=few)
        ix.close()
