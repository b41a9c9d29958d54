"""The operator mirror (`GpuAccelerator`, `GpuResidentIndex`), the collection mirror and the shard
merge, through the C ABI, against the oracle's restatement of the reference's CPU arm
(src/experimental/gpu/common.rs:515-581 arithmetic, gpu/mod.rs operators) and of
Collection::batch_search (src/collection/batch.rs).  Ids bit-exact, distances bit-equal."""
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


@pytest.fixture(scope="module")
def accel(ctx):
    from needle_amd.accelerator import GpuAccelerator

    return GpuAccelerator(ctx=ctx)


def _bits(a):
    return np.asarray(a, dtype=F).view(np.uint32)


@pytest.mark.parametrize("dim", [3, 16, 130, 384])
def test_batch_distance_operators(accel, oracle, dim):
    rng = np.random.default_rng(dim)
    base = rng.standard_normal((777, dim)).astype(F)
    q = rng.standard_normal(dim).astype(F)
    base[5] = 0.0  # zero row: cosine distance 1.0 (gpu/common.rs cosine guard)
    for fn, dt in ((accel.batch_cosine_distance, oracle.ACCEL_COSINE), (accel.batch_euclidean_distance, oracle.ACCEL_EUCLIDEAN),
                   (accel.batch_dot_product, oracle.ACCEL_DOT)):
        got = fn(q, base)
        want = oracle.accel_batch_distance(q, base, dt)
        assert np.array_equal(_bits(got), _bits(want)), (dt, got[:4], want[:4])
    assert accel.batch_cosine_distance(q, np.zeros((0, dim), dtype=F)).size == 0  # gpu/mod.rs:315-317
    import needle_amd as na

    with pytest.raises(na.DimensionMismatch):
        accel.batch_cosine_distance(q, [list(range(dim + 1))])


@pytest.mark.parametrize("dt_name", ["Cosine", "Euclidean", "DotProduct"])
def test_fused_and_batch_fused_search(accel, oracle, dt_name):
    from needle_amd.distance import DistanceType

    dt = DistanceType[dt_name]
    rng = np.random.default_rng(int(dt) + 10)
    base = rng.standard_normal((6000, 70)).astype(F)
    queries = rng.standard_normal((33, 70)).astype(F)
    oi, od, oc = oracle.accel_batch_fused_search(queries, base, 10, int(dt))
    got = accel.batch_fused_search(queries, base, 10, dt)
    assert len(got) == 33
    for i, hits in enumerate(got):
        assert [h[0] for h in hits] == oi[i, : oc[i]].tolist()
        assert np.array_equal(_bits([h[1] for h in hits]), _bits(od[i, : oc[i]]))
    one = accel.fused_search(queries[0], base, 7, dt)
    si, sd = oracle.accel_fused_search(queries[0], base, 7, int(dt))
    assert [h[0] for h in one] == si.tolist() and np.array_equal(_bits([h[1] for h in one]), _bits(sd))
    assert accel.batch_fused_search([], base, 5, dt) == [] and accel.batch_fused_search(queries, [], 5, dt) == []
    big = accel.fused_search(queries[1], base[:4], 10, dt)  # k > n
    assert len(big) == 4


def test_streaming_and_prefiltered_search(accel, oracle):
    from needle_amd.distance import DistanceType

    rng = np.random.default_rng(3)
    base = rng.standard_normal((5000, 48)).astype(F)
    q = rng.standard_normal(48).astype(F)
    for chunk in (64, 1000, 10000):  # the reference merges per-chunk top-k lists: same set as one search
        si, sd = oracle.accel_streaming_search(q, base, 9, oracle.ACCEL_EUCLIDEAN, chunk)
        got = accel.streaming_batch_search(q, base, 9, DistanceType.Euclidean, chunk_size=chunk)
        assert [h[0] for h in got] == si.tolist() and np.array_equal(_bits([h[1] for h in got]), _bits(sd))
    # many chunks (more (distance, id) entries than one device merge takes: folded into a running top k), and a k for
    # which even two lists exceed one merge (served by one resident search): the same hits as the unchunked search
    for kq, chunk in ((10, 8), (400, 500), (2000, 2500)):
        si, sd = oracle.accel_fused_search(q, base, kq, oracle.ACCEL_EUCLIDEAN)
        got = accel.streaming_batch_search(q, base, kq, DistanceType.Euclidean, chunk_size=chunk)
        assert [h[0] for h in got] == si.tolist() and np.array_equal(_bits([h[1] for h in got]), _bits(sd)), (kq, chunk)
    # prefiltered: (original index, vector) pairs, predicate on the original index (gpu/mod.rs:1033-1080)
    pairs = [(1000 + 3 * i, base[i]) for i in range(2000)]
    keep = lambda idx: idx % 2 == 0  # noqa: E731
    got = accel.prefiltered_batch_search(q, pairs, 5, DistanceType.Cosine, keep)
    kept = [(idx, v) for idx, v in pairs if keep(idx)]
    si, sd = oracle.accel_fused_search(q, np.stack([v for _, v in kept]), 5, oracle.ACCEL_COSINE)
    assert [h[0] for h in got] == [kept[int(j)][0] for j in si]
    assert np.array_equal(_bits([h[1] for h in got]), _bits(sd))
    assert accel.prefiltered_batch_search(q, pairs, 5, DistanceType.Cosine, lambda idx: False) == []


def test_kmeans_assign(accel, oracle):
    rng = np.random.default_rng(4)
    vecs = rng.standard_normal((9000, 40)).astype(F)
    cents = rng.standard_normal((300, 40)).astype(F)
    cents[17] = cents[3]  # equal centroids: the first one wins (gpu/mod.rs:660-703 strict <)
    got = accel.kmeans_assign(vecs, cents)
    want = oracle.accel_kmeans_assign(vecs, cents)
    assert got == want.tolist()


def test_merge_topk_device(ctx, oracle):
    import torch

    rng = np.random.default_rng(5)
    lists, q, per, k = 4, 50, 12, 10
    dists = np.sort(rng.random((lists, q, per)).astype(F), axis=2)
    ids = rng.permutation(lists * q * per).reshape(lists, q, per).astype(np.uint64)
    counts = rng.integers(0, per + 1, size=(lists, q)).astype(np.uint32)
    dists[1, :, 0] = dists[0, :, 0]  # cross-shard distance ties: the lower id wins
    dev = torch.device("cuda", 0)
    td = torch.from_numpy(dists).to(dev)
    ti = torch.from_numpy(ids.view(np.int64)).to(dev)
    tc = torch.from_numpy(counts.view(np.int32)).to(dev)
    oi = torch.empty((q, k), dtype=torch.int64, device=dev)
    od = torch.empty((q, k), dtype=torch.float32, device=dev)
    oc = torch.empty((q,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.merge_topk_device(td, ti, tc, lists, q, per, k, oi, od, oc)
    gi, gd, gc = oi.cpu().numpy().view(np.uint64), od.cpu().numpy(), oc.cpu().numpy()
    for j in range(q):
        wi, wd = oracle.merge_topk(dists[:, j, :], ids[:, j, :], counts[:, j], k)
        assert gc[j] == wi.size
        assert np.array_equal(gi[j, : wi.size], wi) and np.array_equal(_bits(gd[j, : wi.size]), _bits(wd))
        assert np.all(gi[j, wi.size:] == U64MAX)


def test_resident_index(ctx, oracle):
    from needle_amd.accelerator import GpuResidentIndex
    from needle_amd.distance import DistanceType
    import needle_amd as na

    rng = np.random.default_rng(6)
    rows = rng.standard_normal((400, 24)).astype(F)
    ix = GpuResidentIndex(24, DistanceType.Euclidean, ctx=ctx)
    assert ix.search(rows[0], 5) == []  # empty index
    ix.add_batch([(10_000 + i, rows[i]) for i in range(400)])
    assert len(ix) == 400 and ix.dirty
    q = rng.standard_normal(24).astype(F)
    si, sd = oracle.accel_fused_search(q, rows, 6, oracle.ACCEL_EUCLIDEAN)
    got = ix.search(q, 6)
    assert not ix.dirty
    assert [h[0] for h in got] == [10_000 + int(j) for j in si] and np.array_equal(_bits([h[1] for h in got]), _bits(sd))
    assert ix.remove(10_000 + int(si[0])) and not ix.remove(5)
    keep = np.ones(400, dtype=bool)
    keep[int(si[0])] = False
    ids_left = np.flatnonzero(keep)
    si2, sd2 = oracle.accel_fused_search(q, rows[keep], 6, oracle.ACCEL_EUCLIDEAN)
    got = ix.search(q, 6)
    assert [h[0] for h in got] == [10_000 + int(ids_left[int(j)]) for j in si2]
    with pytest.raises(na.DimensionMismatch):
        ix.add(1, rows[0][:5])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_collection_batch_search(ctx, oracle, precision):
    import needle_amd as na
    from needle_amd.collection import Collection

    rng = np.random.default_rng(7)
    n, dim = 3000, 64
    rows = rng.standard_normal((n, dim)).astype(F)
    col = Collection("docs", dim, na.DistanceFunction.Cosine, ctx=ctx, precision=precision)
    for i in range(n):
        col.insert(f"doc{i}", rows[i], {"group": i % 5})
    queries = rng.standard_normal((17, dim)).astype(F)
    oi, od, oc = oracle.flat_batch_search(queries, rows, 10, int(na.DistanceFunction.Cosine))
    res = col.batch_search(queries, 10)
    for i, hits in enumerate(res):
        assert [h.id for h in hits] == [f"doc{int(j)}" for j in oi[i, : oc[i]]]
        assert np.array_equal(_bits([h.distance for h in hits]), _bits(od[i, : oc[i]]))
        assert all(h.metadata == {"group": int(h.id[3:]) % 5} for h in hits)
    # deletes are tombstones: the row stays resident and is masked (collection delete + search filter)
    gone = {int(j) for j in oi[:, 0]}
    for j in gone:
        assert col.delete(f"doc{j}")
    assert not col.delete("doc-missing") and len(col) == n - len(gone)
    valid = np.ones(n, dtype=bool)
    valid[list(gone)] = False
    oi, od, oc = oracle.flat_batch_search(queries, rows, 10, int(na.DistanceFunction.Cosine), valid=valid)
    res = col.batch_search(queries, 10)
    for i, hits in enumerate(res):
        assert [h.id for h in hits] == [f"doc{int(j)}" for j in oi[i, : oc[i]]]
    # metadata pre-filter (batch_search_with_filter, batch.rs:108-171)
    filt = (np.arange(n) % 5) == 2
    oi, od, oc = oracle.flat_batch_search(queries, rows, 8, int(na.DistanceFunction.Cosine), valid=valid, filt=filt)
    res = col.batch_search_with_filter(queries, 8, lambda m: m is not None and m["group"] == 2)
    for i, hits in enumerate(res):
        assert [h.id for h in hits] == [f"doc{int(j)}" for j in oi[i, : oc[i]]]
        assert np.array_equal(_bits([h.distance for h in hits]), _bits(od[i, : oc[i]]))
    # validation: the whole batch is rejected before any search runs (batch.rs:40-47)
    bad = queries.copy()
    bad[3, 5] = np.nan
    with pytest.raises(na.InvalidVector, match="NaN at index 5"):
        col.batch_search(bad, 5)
    with pytest.raises(na.DimensionMismatch, match="expected 64, got 10"):
        col.batch_search([list(range(10))], 5)
    with pytest.raises(ValueError):
        col.insert("doc1", rows[1])
    assert col.batch_search(queries, 0) == [[] for _ in range(17)]
    assert col.batch_search([], 5) == []
    assert len(col.search(queries[0], 10_000)) == len(col)  # clamp_k


def test_collection_filter_objects_and_ttl(ctx, oracle):
    """Filter (JSON form, builders, inverted-index path) and lazy expiration produce the row masks; the scan
    then returns exactly the oracle's pre-filtered result (search_methods.rs:205-223, ttl.rs:17-23)."""
    import needle_amd as na
    from needle_amd.collection import Collection
    from needle_amd.metadata import Filter

    rng = np.random.default_rng(21)
    n, dim = 2500, 48
    rows = rng.standard_normal((n, dim)).astype(F)
    now = [5000]
    col = Collection("docs", dim, na.DistanceFunction.Euclidean, ctx=ctx, precision="bf16", clock=lambda: now[0])
    meta = [{"group": i % 7, "tags": ["even" if i % 2 == 0 else "odd"], "info": {"score": i / n}} for i in range(n)]
    for i in range(n):
        col.insert_with_ttl(f"d{i}", rows[i], meta[i] if i % 50 else None, 100 if i % 3 == 0 else None)
    queries = rng.standard_normal((9, dim)).astype(F)
    met = int(na.DistanceFunction.Euclidean)

    def check(res, keep, k):
        oi, od, oc = oracle.flat_batch_search(queries, rows, k, met, valid=keep)
        for i, hits in enumerate(res):
            assert [h.id for h in hits] == [f"d{int(j)}" for j in oi[i, : oc[i]]]
            assert np.array_equal(_bits([h.distance for h in hits]), _bits(od[i, : oc[i]]))

    has_meta = np.array([i % 50 != 0 for i in range(n)])
    grp = np.array([i % 7 for i in range(n)])
    check(col.batch_search_with_filter(queries, 10, Filter.eq("group", 3)), has_meta & (grp == 3), 10)  # inverted index
    check(col.batch_search_with_filter(queries, 10, {"$or": [{"group": {"$in": [1, 2]}}, {"info.score": {"$lt": 0.1}}]}),
          has_meta & (np.isin(grp, [1, 2]) | (np.arange(n) / n < 0.1)), 10)
    check(col.batch_search_with_filter(queries, 5, {"tags": {"$contains": "odd"}, "group": {"$ne": 0}}),
          has_meta & (np.arange(n) % 2 == 1) & (grp != 0), 5)
    check(col.batch_search_with_filter(queries, 5, Filter.exists("group", False)), ~has_meta, 5)
    now[0] = 5100  # every third row has expired: masked lazily, still counted by len()
    live = np.arange(n) % 3 != 0
    assert len(col) == n
    check(col.batch_search(queries, 10), live, 10)
    check(col.batch_search_with_filter(queries, 10, Filter.eq("group", 3)), live & has_meta & (grp == 3), 10)
    assert col.expire_vectors() == int((~live).sum()) and len(col) == int(live.sum())
    check(col.batch_search(queries, 10), live, 10)


@pytest.mark.parametrize("shape", [(4000, 64, (32, 16), 16, 10, 4), (20000, 384, (128, 64), 200, 10, 3),
                                   (3000, 100, (50,), 50, 7, 1), (9, 8, (4, 2), 4, 20, 3), (6000, 256, (64,), 32, 100, 20)])
def test_adaptive_search_two_phase(ctx, oracle, shape):
    """adaptive_search (quantization.rs:890-943) on the device: prefix scan, full-dimension re-rank; ids, both
    distances and the tie order equal the oracle's, per query."""
    from needle_amd.matryoshka import AdaptiveIndex, MatryoshkaTruncation

    n, dim, levels, coarse_dims, k, overfetch = shape
    rng = np.random.default_rng(n + dim)
    # leading dimensions carry most of the signal, as Matryoshka embeddings do
    rows = (rng.standard_normal((n, dim)) * np.linspace(2.0, 0.2, dim)).astype(F)
    if n > 100:
        rows[50:56] = rows[7]            # identical rows: full-distance ties, decided by (coarse distance, row) descending
        rows[60, : dim // 2] = rows[7, : dim // 2]
    trunc = MatryoshkaTruncation(dim, levels)
    queries = np.concatenate([rows[7:8], (rng.standard_normal((23, dim)) * np.linspace(2.0, 0.2, dim)).astype(F)])
    ix = AdaptiveIndex(ctx, rows, trunc)
    ids, full, coarse, counts = ix.search_arrays(queries, coarse_dims, k, overfetch)
    eff = trunc.nearest_level(coarse_dims)
    for i in range(queries.shape[0]):
        oi, of, oc = oracle.adaptive_search(queries[i], rows, eff, k, overfetch)
        c = len(oi)
        assert counts[i] == c == min(k, min(k * max(overfetch, 1), n)), (i, counts[i], c)
        assert np.array_equal(ids[i, :c], oi), (i, ids[i, :c], oi)
        assert np.array_equal(_bits(full[i, :c]), _bits(of)) and np.array_equal(_bits(coarse[i, :c]), _bits(oc)), i
        assert np.all(ids[i, c:] == U64MAX)
    one = ix.adaptive_search(queries[0], coarse_dims, k, overfetch)
    assert one[0].id in (7, 50, 51, 52, 53, 54, 55) if n > 100 else one[0].distance >= 0.0
    assert one[0].distance == 0.0 if n > 100 else True
    import needle_amd as na

    with pytest.raises(na.DimensionMismatch):
        ix.search_arrays(queries[:, : dim - 1], coarse_dims, k, overfetch)
    assert ix.search_arrays(queries[:3], coarse_dims, 0, overfetch)[3].tolist() == [0, 0, 0]
    ix.close()


def test_batch_top_k_reference_kats(accel):
    """The two value-pinning tests of the reference's selection operators, through the device:
    batch_top_k (src/experimental/gpu/mod.rs:1669-1679) and partial_sort_top_k (:1964-1972)."""
    got = accel.batch_top_k([0.5, 0.1, 0.8, 0.3, 0.2], 3)
    assert [h[0] for h in got] == [1, 4, 3]
    assert np.array_equal(_bits([h[1] for h in got]), _bits([0.1, 0.2, 0.3]))
    got = accel.batch_top_k([0.5, 0.1, 0.8, 0.3, 0.2, 0.9, 0.15], 3)
    assert [h[0] for h in got] == [1, 6, 4]
    assert accel.batch_top_k([], 3) == [] and accel.batch_top_k([1.0], 0) == []
    assert [h[0] for h in accel.batch_top_k([2.0, 1.0], 5)] == [1, 0]  # k > n: everything, sorted


@pytest.mark.parametrize("shape", [(1, 100, 10), (3, 5000, 1), (2, 70000, 100), (1, 1_000_003, 10), (4, 9000, 2048),
                                   (1, 300, 300), (2, 4097, 64)])
def test_batch_top_k_against_oracle(ctx, oracle, shape):
    rows, n, k = shape
    rng = np.random.default_rng(rows * 1000 + k)
    s = rng.standard_normal((rows, n)).astype(F)
    s[:, ::7] = np.round(s[:, ::7], 1)        # many exact ties, decided by index
    s[0, 5] = -0.0
    s[0, 3] = 0.0                              # -0.0 == +0.0: index order
    s[-1, n // 2:] = s[-1, n // 2]             # a long run of equal scores across block boundaries
    idx, val, cnt = ctx.batch_top_k(s, k)
    for r in range(rows):
        oi, ov = oracle.top_k(s[r], k)
        assert cnt[r] == oi.size == min(k, n)
        assert np.array_equal(idx[r, : oi.size], oi), (r, idx[r, :8], oi[:8])
        assert np.array_equal(_bits(val[r, : oi.size]), _bits(s[r][oi.astype(np.int64)]))
    inf = np.full((1, 50), np.inf, dtype=F)
    inf[0, 10] = np.nan                        # NaN sorts behind +inf
    idx, val, cnt = ctx.batch_top_k(inf, 50)
    assert idx[0, -1] == 10 and idx[0, :10].tolist() == list(range(10))


def test_normalize_reference_kats(ctx, oracle):
    """normalize (3,4) -> (.6,.8) (src/distance.rs:899-904), zero vector untouched (:1077-1082), and both
    normalisation recipes bit-equal to the scalar restatement on random rows."""
    import needle_amd as na

    v = na.normalized([3.0, 4.0], ctx)
    assert abs(v[0] - 0.6) < 1e-6 and abs(v[1] - 0.8) < 1e-6
    assert np.all(na.normalized([0.0, 0.0, 0.0], ctx) == 0)
    w = np.array([3.0, 4.0], dtype=F)
    na.normalize(w, ctx)
    assert np.array_equal(_bits(w), _bits(v))
    rng = np.random.default_rng(11)
    for dim in (1, 3, 128, 384, 770):
        rows = (rng.standard_normal((257, dim)) * rng.uniform(1e-3, 1e3, (257, 1))).astype(F)
        rows[7] = 0.0
        got = ctx.normalize(rows)
        want = np.stack([oracle.normalized(r) for r in rows])
        assert np.array_equal(_bits(got), _bits(want)), dim
        # the accelerator's recipe: chunk-of-four norm, division (gpu/common.rs:628-635)
        got4 = ctx.normalize(rows, na._lib.ARITH_ACCEL_CHUNK4)
        for i in (0, 7, 100, 256):
            nrm = np.sqrt(F(oracle.accel_batch_distance(rows[i], rows[i:i + 1], oracle.ACCEL_DOT)[0]))
            want4 = rows[i] / nrm if nrm > 0 else rows[i]
            assert np.array_equal(_bits(got4[i]), _bits(want4.astype(F))), (dim, i)


def test_merge_topk_host(ctx, oracle):
    rng = np.random.default_rng(15)
    lists, q, per, k = 3, 20, 8, 6
    dists = np.sort(rng.random((lists, q, per)).astype(F), axis=2)
    ids = rng.permutation(lists * q * per).reshape(lists, q, per).astype(np.uint64)
    counts = rng.integers(0, per + 1, size=(lists, q)).astype(np.uint32)
    gi, gd, gc = ctx.merge_topk(dists, ids, counts, k)
    for j in range(q):
        wi, wd = oracle.merge_topk(dists[:, j, :], ids[:, j, :], counts[:, j], k)
        assert gc[j] == wi.size and np.array_equal(gi[j, : wi.size], wi) and np.array_equal(_bits(gd[j, : wi.size]), _bits(wd))
