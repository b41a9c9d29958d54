"""Planning overrides (needle_hip_set_tuning) change how a search is planned, never what it returns: under every
override the flat and the inverted-file search still answer like the oracle.  The overrides also force the rarely
taken forms of the post-processing kernels (unseeded thresholds: candidate concatenations far larger than the merge
buffer; tiny merge buffers: the list-by-list gather; small re-score blocks; no pruning; a single launch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.set_tuning(reset=1)
    c.close()


def _equal(oracle_out, got):
    oi, od, oc = oracle_out
    ids, dists, counts = got
    assert np.array_equal(counts, oc)
    for i in range(oi.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), i


FLAT_OVERRIDES = [
    dict(debug=2),                          # no threshold seed: every list fills up, the merge takes its input in pieces
    dict(debug=2, merge_keys=512, merge_lists=8),   # ... and list by list
    dict(debug=4),                          # one launch
    dict(debug=16),                         # no pruning before the exact re-score
    dict(stages=(30, 30)),
    dict(stages=(3, 7, 15, 30)),
    dict(splits=7),
    dict(splits=64, pre_tiles=2),
    dict(band=3.0, kseed=40),
    dict(rs_cap=8, rs_dc=32),
    dict(merge_keys=512, merge_lists=4),
    dict(no_dense=1),
]


@pytest.mark.parametrize("k", [10, 100])
def test_flat_results_do_not_depend_on_the_plan(ctx, oracle, k):
    import needle_amd as na

    rng = np.random.default_rng(70 + k)
    n, d, nq = 60_000, 64, 130
    base = rng.standard_normal((n, d)).astype(F)
    base[40_000:40_300] = base[17]           # 300 equal rows: ties across slices
    queries = rng.standard_normal((nq, d)).astype(F)
    queries[0] = base[17]
    valid = rng.random(n) > 0.05
    want = oracle.flat_batch_search(queries, base, k, int(na.DistanceFunction.Euclidean), valid=valid)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean, valid=valid)
    small = na.FlatIndex(ctx, base[:3000], na.DistanceFunction.Euclidean)   # the dense route (no_dense sends it to the lists)
    want_small = oracle.flat_batch_search(queries, base[:3000], k, int(na.DistanceFunction.Euclidean))
    try:
        for over in FLAT_OVERRIDES:
            ctx.set_tuning(reset=1)
            ctx.set_tuning(**over)
            for prec in ("bf16", "fp32"):
                _equal(want, ix.search(queries, k, precision=prec))
            _equal(want_small, small.search(queries, k, precision="bf16"))
    finally:
        ctx.set_tuning(reset=1)
        ix.close()
        small.close()


IVF_OVERRIDES = [dict(ivf_no_seed=1), dict(ivf_seed_rows=16), dict(ivf_seed_rows=256), dict(ivf_split=2), dict(ivf_split=8),
                 dict(no_set_only=1), dict(debug=16), dict(merge_keys=512, merge_lists=4), dict(rs_cap=8, rs_dc=32)]


def test_ivf_results_do_not_depend_on_the_plan(ctx, oracle):
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(71)
    n, d, nlist, nprobe, k = 20_000, 48, 64, 12, 10
    base = rng.standard_normal((n, d)).astype(F)
    queries = rng.standard_normal((90, d)).astype(F)
    cents = oracle.ivf_kmeans(base[:4000], nlist, max_iter=4, seed=3)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    want = oracle.ivf_batch_search(queries, base, cents, offsets, order, nprobe, k)
    ix = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(n), base)
    try:
        for over in IVF_OVERRIDES:
            ctx.set_tuning(reset=1)
            ctx.set_tuning(**over)
            for prec in ("bf16", "fp32"):
                _equal(want, ix.batch_search_arrays(queries, k, precision=prec))
    finally:
        ctx.set_tuning(reset=1)
        ix.close()
