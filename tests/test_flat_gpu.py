"""Parity of the HIP exact flat search against the CPU oracle, through the C ABI.

Bar: ids bit-exact (ties broken by id), distances bit-equal to the oracle's
reference-order arithmetic (stricter than the 1e-4 relative the task allows).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32
U64MAX = np.iinfo(np.uint64).max


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _data(seed, n, d, q, kind="gauss"):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(F), rng.standard_normal((q, d)).astype(F)
    if kind == "unit":
        b = rng.standard_normal((n, d)).astype(F)
        qq = rng.standard_normal((q, d)).astype(F)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        qq /= np.linalg.norm(qq, axis=1, keepdims=True)
        return b.astype(F), qq.astype(F)
    if kind == "uniform":
        return rng.random((n, d), dtype=F), rng.random((q, d), dtype=F)
    raise ValueError(kind)


def _check(oracle, got, base, queries, k, metric, valid=None, filt=None):
    ids, dists, counts = got
    oi, od, oc = oracle.flat_batch_search(queries, base, k, int(metric), valid=valid, filt=filt)
    assert np.array_equal(counts, oc), (counts[:8], oc[:8])
    for i in range(queries.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (i, dists[i, :c], od[i, :c])
        assert np.all(ids[i, c:] == U64MAX) and np.all(np.isinf(dists[i, c:]))


METRICS4 = ["Cosine", "CosineNormalized", "Euclidean", "DotProduct"]


@pytest.mark.parametrize("metric", METRICS4)
@pytest.mark.parametrize("shape", [(1000, 16, 7, 5), (3000, 100, 130, 10), (517, 33, 3, 1)])
def test_small_shapes_all_metrics(ctx, oracle, metric, shape):
    import needle_amd as na

    n, d, q, k = shape
    base, queries = _data(11, n, d, q, "unit" if metric == "CosineNormalized" else "gauss")
    m = na.DistanceFunction[metric]
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, k), base, queries, k, m)
    ix.close()


def test_plumbing_shape_l2_128(ctx, oracle):
    """Scaled-down 'exact flat, 100K x 128 L2, 1K queries, k=10'."""
    import needle_amd as na

    base, queries = _data(1, 40000, 128, 256, "gauss")
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    _check(oracle, ix.search(queries, 10), base, queries, 10, na.DistanceFunction.Euclidean)
    ix.close()


def test_readme_shape_cosine_384(ctx, oracle):
    """Scaled-down 'exact flat, 1M x 384 cosine, 10K queries, k=10' on normalized Gaussian embeddings."""
    import needle_amd as na

    base, queries = _data(2, 30000, 384, 300, "unit")
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    _check(oracle, ix.search(queries, 10), base, queries, 10, na.DistanceFunction.Cosine)
    ix.close()


@pytest.mark.parametrize("metric", METRICS4)
def test_long_slices_seeded_threshold(ctx, oracle, metric):
    """Slices long enough for the threshold-seeding pre-pass, with and without masks."""
    import needle_amd as na

    n, d, q, k = 150_000, 32, 200, 10
    base, queries = _data(21, n, d, q, "unit" if metric == "CosineNormalized" else "gauss")
    m = na.DistanceFunction[metric]
    rng = np.random.default_rng(3)
    valid = rng.random(n) > 0.2
    filt = rng.random(n) < 0.1
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, k), base, queries, k, m)
    ix.set_valid_mask(valid)
    _check(oracle, ix.search(queries, k, filter_mask=filt), base, queries, k, m, valid=valid, filt=filt)
    ix.close()


def test_sorted_base_worst_case_order(ctx, oracle):
    """Rows arranged from farthest to nearest: every tile beats the running threshold."""
    import needle_amd as na

    base, queries = _data(22, 60_000, 24, 8)
    q0 = queries[0]
    order = np.argsort(-((base - q0) ** 2).sum(axis=1))
    base = np.ascontiguousarray(base[order])
    m = na.DistanceFunction.Euclidean
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, 10), base, queries, 10, m)
    ix.close()


def test_gen_vector_inputs(ctx, oracle):
    """The reference's own LCG inputs (src/recall_benchmark.rs:171-180, 237-242)."""
    import needle_amd as na

    base = oracle.gen_vectors(42, 0, 5000, 128)
    queries = oracle.gen_vectors(42 + 999_999, 0, 100, 128)
    for m in (na.DistanceFunction.Cosine, na.DistanceFunction.Euclidean):
        ix = na.FlatIndex(ctx, base, m)
        _check(oracle, ix.search(queries, 10), base, queries, 10, m)
        ix.close()


def test_k_edge_cases(ctx, oracle):
    import needle_amd as na

    base, queries = _data(3, 50, 24, 5)
    m = na.DistanceFunction.Euclidean
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, 100), base, queries, 100, m)  # k > n: clamp_k
    ids, d, c = ix.search(queries, 0)  # k == 0
    assert ids.shape == (5, 0) and c.tolist() == [0] * 5
    _check(oracle, ix.search(queries, 50), base, queries, 50, m)  # k == n
    ix.close()
    empty = na.FlatIndex(ctx, np.zeros((0, 24), dtype=F), m, dim=24)  # empty collection
    ids, d, c = empty.search(queries, 5)
    assert c.tolist() == [0] * 5 and np.all(ids == U64MAX)
    assert len(empty) == 0
    empty.close()


def test_large_k(ctx, oracle):
    import needle_amd as na

    base, queries = _data(4, 5000, 64, 20)
    m = na.DistanceFunction.DotProduct
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, 100), base, queries, 100, m)   # first pass with kprime = 128
    _check(oracle, ix.search(queries, 500), base, queries, 500, m)   # beyond the first pass: exact scan
    ix.close()


def test_ties_broken_by_id(ctx, oracle):
    import needle_amd as na

    base, queries = _data(5, 2000, 32, 4)
    base[100] = base[7]
    base[1500] = base[7]
    base[1999] = base[7]
    queries[0] = base[7]
    for m in (na.DistanceFunction.Euclidean, na.DistanceFunction.Cosine, na.DistanceFunction.DotProduct):
        ix = na.FlatIndex(ctx, base, m)
        got = ix.search(queries, 6)
        _check(oracle, got, base, queries, 6, m)
        if m != na.DistanceFunction.DotProduct:
            assert got[0][0, :4].tolist() == [7, 100, 1500, 1999]
        ix.close()


def test_deleted_and_filter_masks(ctx, oracle):
    import needle_amd as na

    base, queries = _data(6, 4000, 48, 33)
    rng = np.random.default_rng(0)
    valid = rng.random(4000) > 0.3
    filt = (np.arange(4000) % 10) == 0  # 10 % selectivity
    m = na.DistanceFunction.DotProduct
    ix = na.FlatIndex(ctx, base, m, valid=valid)
    _check(oracle, ix.search(queries, 10), base, queries, 10, m, valid=valid)
    _check(oracle, ix.search(queries, 10, filter_mask=filt), base, queries, 10, m, valid=valid, filt=filt)
    _check(oracle, ix.search(queries, 400, filter_mask=filt), base, queries, 400, m, valid=valid, filt=filt)
    ix.set_valid_mask(None)
    _check(oracle, ix.search(queries, 10), base, queries, 10, m)
    none = np.zeros(4000, dtype=bool)
    ids, d, c = ix.search(queries, 10, filter_mask=none)
    assert c.tolist() == [0] * 33
    ix.close()


def test_bound_check_sends_near_duplicates_to_exact_scan(ctx, oracle):
    """More near-identical rows than the first pass keeps: its scores cannot order them, the
    bound check must notice and the exact scan must still return the oracle's answer."""
    import needle_amd as na

    rng = np.random.default_rng(7)
    d = 64
    centre = rng.standard_normal(d).astype(F)
    base = rng.standard_normal((3000, d)).astype(F)
    base[500:800] = centre + (rng.standard_normal((300, d)) * 1e-7).astype(F)
    queries = np.stack([centre, centre * F(1.5), rng.standard_normal(d).astype(F)])
    for m in (na.DistanceFunction.Cosine, na.DistanceFunction.Euclidean, na.DistanceFunction.DotProduct):
        ctx.reset_metrics()
        ix = na.FlatIndex(ctx, base, m)
        _check(oracle, ix.search(queries, 10), base, queries, 10, m)
        ix.close()
        if m != na.DistanceFunction.DotProduct:
            assert ctx.metrics()["fallback_queries"] >= 1


@pytest.mark.parametrize("metric", ["Cosine", "CosineNormalized", "Euclidean", "DotProduct", "Manhattan", "Hamming",
                                    "Chebyshev"])
def test_exact_scan_all_metrics(ctx, oracle, metric):
    import needle_amd as na

    base, queries = _data(8, 1500, 40, 9)
    if metric == "Hamming":
        base = (base > 0.5).astype(F)
        queries = (queries > 0.5).astype(F)
    m = na.DistanceFunction[metric]
    ix = na.FlatIndex(ctx, base, m)
    _check(oracle, ix.search(queries, 12, precision="exact"), base, queries, 12, m)
    _check(oracle, ix.search(queries, 12), base, queries, 12, m)
    ix.close()


def test_zero_vectors_cosine(ctx, oracle):
    import needle_amd as na

    base, queries = _data(9, 600, 20, 4)
    base[3] = 0
    base[77] = 0
    queries[1] = 0  # zero query: every distance is 1.0 -> ids 0..k-1
    m = na.DistanceFunction.Cosine
    ix = na.FlatIndex(ctx, base, m)
    got = ix.search(queries, 8)
    _check(oracle, got, base, queries, 8, m)
    assert got[0][1].tolist() == list(range(8))
    ix.close()


def test_build_from_rows(ctx, oracle):
    import needle_amd as na

    base, queries = _data(10, 300, 12, 3)
    rows = [base[i].copy() for i in range(base.shape[0])]  # Vec<Vec<f32>>
    ix = na.FlatIndex.from_rows(ctx, rows, na.DistanceFunction.Euclidean)
    _check(oracle, ix.search(queries, 4), base, queries, 4, na.DistanceFunction.Euclidean)
    ix.close()


def test_errors(ctx):
    import needle_amd as na

    base, queries = _data(12, 100, 8, 2)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    with pytest.raises(na.DimensionMismatch):  # tests/edge_cases.rs: dimension mismatch rejects the batch
        ix.search(np.zeros((2, 7), dtype=F), 3)
    bad = queries.copy()
    bad[1, 3] = np.nan
    with pytest.raises(na.InvalidVector):  # NaN anywhere rejects the whole batch
        ix.search(bad, 3)
    bad[1, 3] = np.inf
    with pytest.raises(na.InvalidVector):
        ix.search(bad, 3)
    ix.close()
    nb = base.copy()
    nb[5, 1] = np.nan
    with pytest.raises(na.InvalidVector):  # validate_insert_input
        na.FlatIndex(ctx, nb, na.DistanceFunction.Cosine)


def test_device_resident_search(ctx, oracle):
    import torch

    import needle_amd as na

    base, queries = _data(13, 6000, 96, 150, "unit")
    dev = torch.device("cuda:0")
    tb = torch.from_numpy(base).to(dev)
    tq = torch.from_numpy(queries).to(dev)
    torch.cuda.synchronize()
    m = na.DistanceFunction.Cosine
    ix = na.FlatIndex(ctx, tb, m)
    k = 10
    ids = torch.empty((150, k), dtype=torch.int64, device=dev)
    dists = torch.empty((150, k), dtype=torch.float32, device=dev)
    counts = torch.empty((150,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ix.search_device(tq, k, ids, dists, counts)
    got = (ids.cpu().numpy().view(np.uint64), dists.cpu().numpy(), counts.cpu().numpy().view(np.uint32))
    _check(oracle, got, base, queries, k, m)
    ix.close()


def test_k_beyond_the_in_lds_ranking(ctx, oracle):
    """k in the thousands (clamp_k allows any k <= len): the exact scan sorts its selection in global memory."""
    import needle_amd as na

    base, queries = _data(9, 6000, 20, 5)
    m = na.DistanceFunction.Euclidean
    valid = np.ones(6000, dtype=bool)
    valid[::7] = False
    ix = na.FlatIndex(ctx, base, m, valid=valid)
    for k in (2049, 4000, 6000):
        _check(oracle, ix.search(queries, k), base, queries, k, m, valid=valid)
    ix.close()


def test_handles_outliving_their_context(oracle):
    """Destroying an index after its context is shut down, or twice, is harmless."""
    import needle_amd as na

    c = na.Context(0)
    base, queries = _data(10, 500, 8, 2)
    ix = na.FlatIndex(c, base, na.DistanceFunction.Cosine)
    ix2 = na.FlatIndex(c, base, na.DistanceFunction.Euclidean)
    ix2.close()
    ix2.close()
    c.close()  # takes ix along
    ix.close()
    c.close()
