"""Parity of the quantized (ADC) scans against the oracle's restatement of
src/indexing/quantization.rs, through the C ABI: ids bit-exact, distances bit-equal."""
import os

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


def _cmp(got, want):
    ids, dists, counts = got
    oi, od, oc = want
    assert np.array_equal(counts, oc), (counts[:8], oc[:8])
    for i in range(ids.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (i, dists[i, :c], od[i, :c])
        assert np.all(ids[i, c:] == U64MAX)


@pytest.mark.parametrize("shape", [(5000, 32, 4, 10), (20000, 96, 12, 10), (3000, 30, 5, 7), (70000, 384, 48, 10),
                                   # 20 and 33 subspaces: a last table slice of 4 and of 1; fewer rows than a lane group; k = 256
                                   (4001, 40, 20, 10), (2500, 66, 33, 3), (13, 32, 16, 5), (30000, 68, 17, 256)])
def test_pq_table_scan(ctx, oracle, shape):
    from needle_amd.quantization import PqIndex, ProductQuantizer

    n, dim, m, k = shape
    rng = np.random.default_rng(n)
    base = rng.standard_normal((n, dim)).astype(F)
    opq = oracle.ProductQuantizer.train(base[:600], m, seed=3)  # seeded codebooks (reference: thread_rng)
    pq = ProductQuantizer(opq.codebooks)
    codes = opq.encode(base)
    assert np.array_equal(pq.encode(base[:200]), codes[:200])  # host mirror of encode == oracle
    queries = rng.standard_normal((37, dim)).astype(F)
    ix = PqIndex(ctx, pq, codes)
    want = opq.batch_search(queries, codes, k)
    _cmp(ix.search(queries, k), want)
    _cmp(ix.search(queries, k, precision="bf16"), want)  # decoded rows on the matrix cores, table arithmetic for the hits
    # the scanned distance equals distance_with_table AND asymmetric_distance within 1e-5 (quantization.rs:1053-1081)
    ids, d, _ = ix.search(queries[:2], 3)
    for qi in range(2):
        for j in range(3):
            assert abs(d[qi, j] - opq.asymmetric_distance(queries[qi], codes[int(ids[qi, j])])) < 1e-5 * max(1, d[qi, j])
    ix.close()


def test_pq_masks_ties_and_edges(ctx, oracle):
    from needle_amd.quantization import PqIndex, ProductQuantizer

    rng = np.random.default_rng(5)
    n, dim, m = 9000, 16, 4
    base = rng.standard_normal((n, dim)).astype(F)
    opq = oracle.ProductQuantizer.train(base[:300], m, seed=1)
    codes = opq.encode(base)
    codes[4000:4100] = codes[17]  # identical codes: exact ties, resolved by row id
    queries = rng.standard_normal((10, dim)).astype(F)
    queries[0] = base[17]
    valid = rng.random(n) > 0.25
    filt = (np.arange(n) % 10) == 0
    ix = PqIndex(ctx, ProductQuantizer(opq.codebooks), codes, valid=valid)
    _cmp(ix.search(queries, 12), opq.batch_search(queries, codes, 12, valid=valid))
    _cmp(ix.search(queries, 12, filter_mask=filt), opq.batch_search(queries, codes, 12, valid=valid & filt))
    _cmp(ix.search(queries, 256), opq.batch_search(queries, codes, 256, valid=valid))
    for kk in (12, 100, 256):  # bf16: the block of identical codes ties at the top; k = 256 exceeds the candidate budget -> table scan
        _cmp(ix.search(queries, kk, precision="bf16"), opq.batch_search(queries, codes, kk, valid=valid))
    _cmp(ix.search(queries, 12, precision="bf16", filter_mask=filt), opq.batch_search(queries, codes, 12, valid=valid & filt))
    for kk, prec in ((300, "fp32"), (700, "bf16")):  # k above the table scan's per-list budget: every row scored in the table's arithmetic
        _cmp(ix.search(queries, kk, precision=prec), opq.batch_search(queries, codes, kk, valid=valid))
    _cmp(ix.search(queries, 300, filter_mask=filt), opq.batch_search(queries, codes, 300, valid=valid & filt))
    ids, _, counts = ix.search(queries, 0)
    assert ids.shape == (10, 0) and counts.tolist() == [0] * 10
    ix.close()
    small = PqIndex(ctx, ProductQuantizer(opq.codebooks), codes[:5])
    _cmp(small.search(queries, 10), opq.batch_search(queries, codes[:5], 10))  # k > n
    _cmp(small.search(queries, 10, precision="bf16"), opq.batch_search(queries, codes[:5], 10))
    small.close()


def test_pq_table_scan_seeded_thresholds(ctx, oracle):
    """The table scan's lists start at the k-th best score of a sample of the rows (8 places spread over the base).  Forced on
    a base small enough for the oracle: rows tying the seed, masks that empty most of the sample, a filter that leaves fewer
    than k live rows in it, and the unseeded plan all give the oracle's answer."""
    from needle_amd.quantization import PqIndex, ProductQuantizer

    rng = np.random.default_rng(77)
    n, dim, m, k = 90_000, 64, 8, 10
    base = rng.standard_normal((n, dim)).astype(F)
    opq = oracle.ProductQuantizer.train(base[:500], m, seed=2)
    codes = opq.encode(base)
    queries = rng.standard_normal((21, dim)).astype(F)
    queries[0] = base[11]
    # rows with the codes of row 11 inside the sampled blocks (ties with the seed itself) and far outside them
    codes[[3, 4000, 11_300, 50_001, 89_999]] = codes[11]
    valid = np.ones(n, dtype=bool)
    valid[:4096:2] = False
    valid[45_000:46_000] = False
    few = np.zeros(n, dtype=bool)
    few[[5, 60_000, 60_001, 75_123]] = True  # fewer than k admitted rows: no seed, every admitted row returned
    sparse = rng.random(n) < 0.02
    ix = PqIndex(ctx, ProductQuantizer(opq.codebooks), codes, valid=valid)
    try:
        for tune in (dict(pre_tiles=1), dict(pre_tiles=1, splits=4), dict(debug=2)):
            ctx.set_tuning(reset=1)
            ctx.set_tuning(**tune)
            _cmp(ix.search(queries, k), opq.batch_search(queries, codes, k, valid=valid))
            _cmp(ix.search(queries, 1), opq.batch_search(queries, codes, 1, valid=valid))
            _cmp(ix.search(queries, 100), opq.batch_search(queries, codes, 100, valid=valid))
            _cmp(ix.search(queries[:6], 256), opq.batch_search(queries[:6], codes, 256, valid=valid))  # lists of 576 entries
            _cmp(ix.search(queries, k, filter_mask=few), opq.batch_search(queries, codes, k, valid=valid & few))
            _cmp(ix.search(queries, k, filter_mask=sparse), opq.batch_search(queries, codes, k, valid=valid & sparse))
    finally:
        ctx.set_tuning(reset=1)
        ix.close()


@pytest.mark.parametrize("shape", [(6000, 32, 10), (30000, 100, 10), (20000, 384, 10), (900, 7, 3)])
def test_sq8_asymmetric_scan(ctx, oracle, shape):
    from needle_amd.quantization import ScalarQuantizer, Sq8Index

    n, dim, k = shape
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(F)
    osq = oracle.ScalarQuantizer.train(base[: max(50, n // 10)])
    sq = ScalarQuantizer.train(base[: max(50, n // 10)])
    assert np.array_equal(sq.min_vals, osq.min_vals) and np.array_equal(sq.scale, osq.scale)
    codes = osq.quantize(base)
    assert np.array_equal(sq.quantize(base[:500]), codes[:500])  # incl. clamping of out-of-range values
    queries = rng.standard_normal((41, dim)).astype(F)
    ix = Sq8Index(ctx, sq, codes)
    _cmp(ix.search(queries, k), osq.batch_search(queries, codes, k))
    _cmp(ix.search(queries[:5], k, precision="exact"), osq.batch_search(queries[:5], codes, k))
    _cmp(ix.search(queries, k, precision="bf16"), osq.batch_search(queries, codes, k))  # bf16 query operand, exact codes
    ix.close()


def test_sq8_masks_duplicates_and_constant_dims(ctx, oracle):
    from needle_amd.quantization import ScalarQuantizer, Sq8Index

    rng = np.random.default_rng(8)
    n, dim = 8000, 24
    base = rng.random((n, dim), dtype=F)
    base[:, 5] = 0.25  # constant dimension: scale falls back to 1.0 (quantization.rs:1125-1136)
    osq = oracle.ScalarQuantizer.train(base)
    assert osq.scale[5] == 1.0
    codes = osq.quantize(base)
    codes[3000:3300] = codes[11]  # identical codes: more ties than the first pass keeps -> exact scan
    queries = rng.random((6, dim), dtype=F)
    queries[0] = base[11]
    valid = rng.random(n) > 0.3
    sq = ScalarQuantizer(osq.min_vals, osq.max_vals, osq.scale)
    ix = Sq8Index(ctx, sq, codes, valid=valid)
    ctx.reset_metrics()
    _cmp(ix.search(queries, 10), osq.batch_search(queries, codes, 10, valid=valid))
    assert ctx.metrics()["fallback_queries"] >= 1
    filt = (np.arange(n) % 7) == 0
    _cmp(ix.search(queries, 10, filter_mask=filt), osq.batch_search(queries, codes, 10, valid=valid & filt))
    ctx.reset_metrics()
    _cmp(ix.search(queries, 10, precision="bf16"), osq.batch_search(queries, codes, 10, valid=valid))
    m2 = ctx.metrics()
    assert m2["refined_queries"] + m2["fallback_queries"] >= 1  # the duplicate cluster goes down the refinement chain
    ix.close()


@pytest.mark.parametrize("shape", [(5000, 64, 10), (30000, 384, 10), (2000, 37, 7), (70000, 100, 100), (3000, 1000, 5)])
def test_binary_hamming_scan(ctx, oracle, shape):
    """BinaryQuantizer codes scanned by Hamming distance: (distance, row id) order of the reference's stable
    sort (compression_advisor.rs:658-671); ties are the rule with integer distances."""
    from needle_amd.quantization import BinaryIndex, BinaryQuantizer

    n, dim, k = shape
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(F)
    obq = oracle.BinaryQuantizer.train(base[: max(50, n // 10)])
    bq = BinaryQuantizer.train(base[: max(50, n // 10)])
    assert np.array_equal(bq.thresholds.view(np.uint32), obq.thresholds.view(np.uint32))
    codes = obq.quantize(base)
    assert np.array_equal(bq.quantize(base[:300]), codes[:300])
    assert np.array_equal(bq.quantize_device(ctx, base[:300]), codes[:300])  # device encoder
    queries = rng.standard_normal((37, dim)).astype(F)
    ix = BinaryIndex(ctx, bq, codes)
    want = obq.batch_search(queries, codes, k)
    _cmp(ix.search(queries, k), want)  # few rows at the threshold distance: one collecting scan
    ctx.set_tuning(binary_two_scans=1)  # the ranked two-scan path on the same data
    try:
        _cmp(ix.search(queries, k), want)
    finally:
        ctx.set_tuning(binary_two_scans=0)
    ix.close()


def test_binary_masks_ties_and_edges(ctx, oracle):
    from needle_amd.quantization import BinaryIndex, BinaryQuantizer

    rng = np.random.default_rng(9)
    n, dim = 20000, 24  # 24 bits: at most 25 distinct distances, thousands of ties at the k-th
    base = rng.standard_normal((n, dim)).astype(F)
    obq = oracle.BinaryQuantizer.train(base)
    codes = obq.quantize(base)
    codes[5000:5200] = codes[3]
    queries = rng.standard_normal((21, dim)).astype(F)
    queries[0] = base[3]
    valid = rng.random(n) > 0.25
    filt = (np.arange(n) % 10) == 0
    bq = BinaryQuantizer(obq.thresholds)
    ix = BinaryIndex(ctx, bq, codes, valid=valid)
    for k in (1, 12, 256, 1000):
        _cmp(ix.search(queries, k), obq.batch_search(queries, codes, k, valid=valid))
    _cmp(ix.search(queries, 12, filter_mask=filt), obq.batch_search(queries, codes, 12, valid=valid & filt))
    none = np.zeros(n, dtype=bool)
    assert ix.search(queries, 5, filter_mask=none)[2].tolist() == [0] * 21
    ids, _, counts = ix.search(queries, 0)
    assert ids.shape == (21, 0) and counts.tolist() == [0] * 21
    ix.close()
    small = BinaryIndex(ctx, bq, codes[:5])
    _cmp(small.search(queries, 10), obq.batch_search(queries, codes[:5], 10))  # k > n
    small.close()


@pytest.mark.parametrize("shape", [(3000, 32, 4), (5000, 96, 12), (200, 24, 3), (1500, 30, 2), (1200, 384, 2), (900, 200, 2)])
def test_pq_training_and_encoding_on_the_device(ctx, oracle, shape):
    """needle_hip_pq_train / needle_hip_pq_encode against the oracle's seeded restatement of
    ProductQuantizer::{train, encode} (quantization.rs:330-446, kmeans :597-711): same draws -> bit-equal
    codebooks; codes equal for every row."""
    from needle_amd.quantization import ProductQuantizer

    n, dim, m = shape
    rng = np.random.default_rng(n + m)
    data = rng.standard_normal((n, dim)).astype(F)
    data[40:70] = data[3]  # repeated rows: zero seeding weights, possible empty clusters
    for seed in (1, 99):
        opq = oracle.ProductQuantizer.train(data, m, seed=seed)
        pq = ProductQuantizer.train_device(ctx, data, m, seed=seed)
        assert np.array_equal(pq.centroids_per_subspace, opq.ncent)
        for s in range(m):
            k = int(opq.ncent[s])
            assert np.array_equal(pq.codebooks[s, :k].view(np.uint32), opq.codebooks[s, :k].view(np.uint32)), s
    probe = rng.standard_normal((777, dim)).astype(F)
    want = opq.encode(probe)
    assert np.array_equal(pq.encode_device(ctx, probe), want)
    assert np.array_equal(pq.encode(probe[:100]), want[:100])
