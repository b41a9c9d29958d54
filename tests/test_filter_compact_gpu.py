"""Filtered flat searches whose masks admit a small part of the rows run over a gathered copy of the admitted rows (work
proportional to the rows that can be returned).  The answers are those of the scan over the whole base -- the oracle's,
bit for bit, ties to the lower row number -- whatever share is admitted, and the metrics say which route a search took."""
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


def _equal(want, got):
    oi, od, oc = want
    ids, dists, counts = got
    assert np.array_equal(counts, oc)
    for i in range(oi.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), i
        assert np.all(ids[i, c:] == np.iinfo(np.uint64).max)


@pytest.mark.parametrize("metric", ["Cosine", "CosineNormalized", "Euclidean", "DotProduct"])
def test_admitted_share_decides_the_route_not_the_answer(ctx, oracle, metric):
    import needle_amd as na

    rng = np.random.default_rng(len(metric))
    n, d, nq = 100_000, 96, 70
    base = rng.standard_normal((n, d)).astype(F)
    queries = rng.standard_normal((nq, d)).astype(F)
    if metric == "CosineNormalized":
        base, queries = oracle.normalize_rows(base), oracle.normalize_rows(queries)
    base[60_000:60_040] = base[123]        # equal rows inside and outside the filter: ties by row number
    queries[0] = base[123]
    valid = rng.random(n) > 0.03           # deleted rows
    m = na.DistanceFunction[metric]
    ix = na.FlatIndex(ctx, base, m, valid=valid)
    for share, compacted in ((0.002, True), (0.1, True), (0.45, True), (0.6, False)):
        filt = rng.random(n) < share
        filt[123] = filt[60_003] = filt[60_020] = True
        live = int(np.count_nonzero(filt & valid))
        for prec in ("bf16", "fp32"):
            for k in (10, 300):            # 300: beyond the candidate budget, the exact scan (of the gathered rows)
                ctx.reset_metrics()
                got = ix.search(queries[: (nq if k == 10 else 6)], k, precision=prec, filter_mask=filt)
                mm = ctx.metrics()
                want = oracle.flat_batch_search(queries[: (nq if k == 10 else 6)], base, k, int(m), valid=valid, filt=filt)
                _equal(want, got)
                assert mm["compacted_searches"] == (1 if compacted else 0), (share, prec, k)
                assert mm["compacted_rows"] == (live if compacted else 0)
    none = np.zeros(n, dtype=bool)
    got = ix.search(queries[:5], 10, precision="bf16", filter_mask=none)
    assert np.all(got[2] == 0)
    # the same filter after the index has grown
    more = rng.standard_normal((30_000, d)).astype(F)
    if metric == "CosineNormalized":
        more = oracle.normalize_rows(more)
    ix.append(more)
    all_rows = np.concatenate([base, more])
    v2 = np.concatenate([valid, np.ones(30_000, dtype=bool)])
    filt = rng.random(n + 30_000) < 0.05
    ctx.reset_metrics()
    _equal(oracle.flat_batch_search(queries, all_rows, 10, int(m), valid=v2, filt=filt),
           ix.search(queries, 10, precision="bf16", filter_mask=filt))
    assert ctx.metrics()["compacted_searches"] == 1
    ix.close()


@pytest.mark.parametrize("dim,n,live_rows", [(7, 1000, 3), (33, 5000, 129), (64, 9000, 1), (384, 3000, 640), (130, 70_000, 9000)])
def test_small_and_ragged_shapes(ctx, oracle, dim, n, live_rows):
    import needle_amd as na

    rng = np.random.default_rng(dim + n)
    base = rng.standard_normal((n, dim)).astype(F)
    queries = rng.standard_normal((17, dim)).astype(F)
    filt = np.zeros(n, dtype=bool)
    filt[rng.choice(n, live_rows, replace=False)] = True
    filt[n - 1] = True                      # the last row: the mask's last word is partial
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    ctx.set_tuning(compact_min_rows=1)
    try:
        for prec in ("bf16", "fp32"):
            for k in (1, 10):
                ctx.reset_metrics()
                _equal(oracle.flat_batch_search(queries, base, k, int(na.DistanceFunction.Euclidean), filt=filt),
                       ix.search(queries, k, precision=prec, filter_mask=filt))
                assert ctx.metrics()["compacted_searches"] == 1
        ctx.set_tuning(no_compact=1)
        ctx.reset_metrics()
        _equal(oracle.flat_batch_search(queries, base, 10, int(na.DistanceFunction.Euclidean), filt=filt),
               ix.search(queries, 10, precision="bf16", filter_mask=filt))
        assert ctx.metrics()["compacted_searches"] == 0
    finally:
        ctx.set_tuning(reset=1)
        ix.close()


def test_device_resident_filter_and_share_threshold(ctx, oracle):
    import torch

    import needle_amd as na

    rng = np.random.default_rng(9)
    n, d, nq, k = 80_000, 64, 40, 10
    base = rng.standard_normal((n, d)).astype(F)
    queries = rng.standard_normal((nq, d)).astype(F)
    filt = rng.random(n) < 0.3
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.DotProduct)
    words = np.packbits(filt, bitorder="little")
    words = np.concatenate([words, np.zeros((-words.size) % 4, dtype=np.uint8)]).view(np.uint32)
    dq = torch.from_numpy(queries).to("cuda:0")
    df = torch.from_numpy(words.view(np.int32).copy()).to("cuda:0")
    ids = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    dists = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
    counts = torch.empty(nq, dtype=torch.int32, device="cuda:0")
    want = oracle.flat_batch_search(queries, base, k, int(na.DistanceFunction.DotProduct), filt=filt)
    try:
        for pct, compacted in ((0, True), (20, False), (35, True)):
            ctx.set_tuning(compact_pct=pct)
            ctx.reset_metrics()
            ix.search_device(dq, k, ids, dists, counts, precision="bf16", d_filter=df)
            torch.cuda.synchronize()
            got = (ids.cpu().numpy().view(np.uint64), dists.cpu().numpy(), counts.cpu().numpy().view(np.uint32))
            _equal(want, got)
            assert ctx.metrics()["compacted_searches"] == (1 if compacted else 0), pct
    finally:
        ctx.set_tuning(reset=1)
        ix.close()


def test_concurrent_filtered_searches_keep_their_own_gathered_rows(ctx, oracle):
    """Four callers search one index at once, each under its own filter: every search slot gathers its own copy."""
    import threading

    import needle_amd as na

    rng = np.random.default_rng(21)
    n, d, nq, k = 90_000, 64, 50, 10
    base = rng.standard_normal((n, d)).astype(F)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    jobs = []
    for t in range(4):
        queries = rng.standard_normal((nq, d)).astype(F)
        filt = rng.random(n) < (0.01, 0.08, 0.3, 0.7)[t]
        jobs.append((queries, filt, oracle.flat_batch_search(queries, base, k, int(na.DistanceFunction.Euclidean), filt=filt)))
    errors = []

    def worker(job):
        queries, filt, want = job
        try:
            for _ in range(6):
                _equal(want, ix.search(queries, k, precision="bf16", filter_mask=filt))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ctx.reset_metrics()
    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]
    assert ctx.metrics()["compacted_searches"] == 18   # three of the four filters admit less than half of the rows
    ix.close()
