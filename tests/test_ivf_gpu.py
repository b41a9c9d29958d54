"""Parity of the device IVF index (assignment, probe lists, list scan) against the CPU oracle's
restatement of src/indexing/ivf.rs, through the C ABI.  Bar: identical assignment, identical probe
lists, ids bit-exact and distances bit-equal."""
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


def _setup(oracle, ctx, n, d, nlist, nprobe, seed, kind="uniform"):
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(seed)
    base = rng.random((n, d), dtype=F) if kind == "uniform" else rng.standard_normal((n, d)).astype(F)
    cents = oracle.ivf_kmeans(base[: min(n, 20 * nlist)], nlist, max_iter=10, seed=seed)
    ix = IvfIndex(d, IvfConfig.new(cents.shape[0]).with_nprobe(nprobe), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(n), base)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, cents.shape[0])
    return base, cents, ix, assign, offsets, order


def _check_search(oracle, ix, base, cents, offsets, order, queries, nprobe, k, **kw):
    ids, dists, counts = ix.batch_search_arrays(queries, k, **kw)
    oi, od, oc = oracle.ivf_batch_search(queries, base, cents, offsets, order, nprobe, k)
    assert np.array_equal(counts, oc), (counts[:8], oc[:8])
    for i in range(queries.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (i, dists[i, :c], od[i, :c])


@pytest.mark.parametrize("shape", [(3000, 32, 16, 4, 10), (20000, 64, 64, 8, 10), (5000, 100, 40, 40, 5),
                                   (4000, 1000, 24, 6, 10), (3000, 1536, 16, 5, 10)])  # (the last two: the widest bf16 list scans)
def test_assignment_probe_and_search(ctx, oracle, shape):
    n, d, nlist, nprobe, k = shape
    base, cents, ix, assign, offsets, order = _setup(oracle, ctx, n, d, nlist, nprobe, seed=n)
    # insert(): nearest centroid of every row (ivf.rs:453-498)
    got_assign = ctx.kmeans_assign(base, cents, 0)
    assert np.array_equal(got_assign, assign)
    assert np.array_equal(ix.cluster_sizes(), np.diff(offsets).astype(np.uint32))
    rng = np.random.default_rng(1)
    queries = np.concatenate([base[:5], rng.random((60, d), dtype=F)])
    # identical probe lists (ivf.rs:521-533)
    probes = ix.probe(queries)
    for i in range(queries.shape[0]):
        assert np.array_equal(probes[i], oracle.ivf_probe(queries[i], cents, nprobe)), i
    _check_search(oracle, ix, base, cents, offsets, order, queries, nprobe, k)
    _check_search(oracle, ix, base, cents, offsets, order, queries, nprobe, k, precision="bf16")  # bf16 list scan, same hits
    ix.close()


def test_hub_lists_are_cut_into_runs(ctx, oracle):
    """Gaussian rows around randomly drawn centroids: a few lists hold most rows and span many tile runs."""
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(12)
    n, d, nlist, nprobe = 60000, 48, 24, 6
    base = rng.standard_normal((n, d)).astype(F)
    cents = np.ascontiguousarray(base[rng.choice(n, nlist, replace=False)] * F(3.0))
    cents[:3] = base[:3] * F(0.1)  # three centroids near the origin attract almost everything
    ix = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(n), base)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    assert np.diff(offsets).max() > 8 * 1024  # several runs of 8 tiles
    queries = rng.standard_normal((300, d)).astype(F)
    for precision in ("fp32", "bf16"):
        _check_search(oracle, ix, base, cents, offsets, order, queries, nprobe, 10, precision=precision)
    ix.close()


def test_reference_kat_self_query_found(ctx, oracle):  # ivf.rs:729-765: nprobe = nlist, k = n
    base, cents, ix, _, offsets, order = _setup(oracle, ctx, 100, 32, 4, 4, seed=7)
    res = ix.search(base[0], 100)
    assert len(res) == 100 and res[0][1] < 1.0
    hit = [r for r in res if r[0] == 0]
    assert hit and hit[0][1] < 0.1
    _check_search(oracle, ix, base, cents, offsets, order, base[:3], 4, 100)
    ix.close()


def test_all_lists_probed_equals_flat(ctx, oracle):
    base, cents, ix, _, offsets, order = _setup(oracle, ctx, 4000, 24, 8, 8, seed=3, kind="gauss")
    rng = np.random.default_rng(2)
    queries = rng.standard_normal((20, 24)).astype(F)
    ids, dists, counts = ix.batch_search_arrays(queries, 7)
    fi, fd, fc = oracle.flat_batch_search(queries, base, 7, oracle.EUCLIDEAN)
    assert np.array_equal(ids, fi) and np.array_equal(dists.view(np.uint32), fd.view(np.uint32))
    ix.close()


def test_edges_k_nprobe_empty_lists(ctx, oracle):
    from needle_amd.ivf import IvfConfig, IvfIndex
    import needle_amd as na

    base, cents, ix, _, offsets, order = _setup(oracle, ctx, 600, 16, 6, 2, seed=5)
    rng = np.random.default_rng(4)
    queries = rng.random((9, 16), dtype=F)
    ids, _, counts = ix.batch_search_arrays(queries, 0)  # k == 0 -> empty (ivf.rs:852-866)
    assert ids.shape == (9, 0) and counts.tolist() == [0] * 9
    _check_search(oracle, ix, base, cents, offsets, order, queries, 2, 500)  # k above the probed rows
    with pytest.raises(na.DimensionMismatch):  # ivf.rs:514-519
        ix.batch_search_arrays(np.zeros((1, 15), dtype=F), 3)
    ix.close()
    # far-away centroids own no rows: empty lists are probed and contribute nothing
    cents2 = np.concatenate([cents, np.full((3, 16), 50.0, dtype=F)])
    ix2 = IvfIndex(16, IvfConfig.new(9).with_nprobe(9), ctx=ctx)
    ix2.set_centroids(cents2)
    ix2.add(range(600), base)
    a2 = oracle.ivf_assign(base, cents2)
    off2, ord2 = oracle.ivf_lists_from_assignment(a2, 9)
    _check_search(oracle, ix2, base, cents2, off2, ord2, queries, 9, 10)
    ix2.close()
    empty = IvfIndex(16, IvfConfig.new(6), ctx=ctx)  # NotTrained (ivf.rs:808-850)
    with pytest.raises(na.NeedleHipError):
        empty.insert(0, base[0])
    with pytest.raises(na.NeedleHipError):
        empty.search(base[0], 3)


def test_near_duplicates_take_the_exact_list_scan(ctx, oracle):
    rng = np.random.default_rng(9)
    d = 32
    base = rng.random((4000, d), dtype=F)
    centre = rng.random(d, dtype=F)
    base[1000:1200] = centre + (rng.standard_normal((200, d)) * 1e-7).astype(F)
    cents = oracle.ivf_kmeans(base[:800], 8, max_iter=10, seed=2)
    from needle_amd.ivf import IvfConfig, IvfIndex

    ix = IvfIndex(d, IvfConfig.new(8).with_nprobe(3), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(4000), base)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, 8)
    queries = np.stack([centre, rng.random(d, dtype=F)])
    ctx.reset_metrics()
    _check_search(oracle, ix, base, cents, offsets, order, queries, 3, 10)
    assert ctx.metrics()["fallback_queries"] >= 1
    _check_search(oracle, ix, base, cents, offsets, order, queries, 3, 10, precision="exact")
    ix.close()


def test_gpu_training_produces_usable_centroids(ctx, oracle):
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(11)
    blobs = np.concatenate([rng.normal(c, 0.05, size=(200, 8)) for c in (0.0, 1.0, 2.0, 3.0)]).astype(F)
    ix = IvfIndex(8, IvfConfig.new(4).with_nprobe(1).with_max_iterations(20), ctx=ctx)
    ix.train(blobs)
    assert ix.is_trained() and ix.centroids.shape == (4, 8)
    ix.add(range(800), blobs)
    sizes = ix.cluster_sizes()
    assert int(sizes.sum()) == 800  # ivf.rs:788-805
    res = ix.search(blobs[5], 3)
    assert res[0][0] == 5 and res[0][1] == 0.0
    ix.close()


@pytest.mark.parametrize("shape", [(2000, 16, 12, 100), (6000, 40, 64, 8), (300, 8, 400, 5)])
def test_training_matches_the_seeded_cpu_kmeans(ctx, oracle, shape):
    """needle_hip_ivf_train against the oracle's restatement of IvfIndex::kmeans (ivf.rs:330-450) with the same
    splitmix64 draws: k-means++ picks, assignments, row-order cluster sums and the stop rules all agree, so the
    centroids are bit-equal."""
    from needle_amd.ivf import IvfConfig, IvfIndex

    n, d, nlist, iters = shape
    rng = np.random.default_rng(n)
    blobs = rng.standard_normal((8, d)).astype(F) * F(4.0)
    data = (blobs[rng.integers(0, 8, n)] + rng.standard_normal((n, d)).astype(F)).astype(F)
    data[50:60] = data[7]  # duplicates: zero-distance rows in the seeding weights
    for seed in (1, 12345):
        want = oracle.ivf_kmeans(data, nlist, max_iter=iters, tol=1e-4, seed=seed)
        ix = IvfIndex(d, IvfConfig.new(nlist).with_max_iterations(iters), ctx=ctx, seed=seed)
        ix.train(data)
        # capped at n (ivf.rs:909-921), and at the number of distinct rows (seeding stops when none is left)
        assert ix.centroids.shape == want.shape == (min(nlist, n - 10), d)
        assert np.array_equal(ix.centroids.view(np.uint32), want.view(np.uint32))
    ix.add(range(n), data)
    assert int(ix.cluster_sizes().sum()) == n  # ivf.rs:788-805
    with pytest.raises(ValueError):
        ix.train(np.zeros((0, d), dtype=F))
    ix.close()


def test_ivf_under_row_masks(ctx, oracle):
    """Deleted rows (valid mask) and a search-time filter: the lists the oracle scans hold the live rows only."""
    import ctypes as C

    from needle_amd import _lib
    from needle_amd.index import _mask_words

    n, d, nlist, nprobe, k = 30000, 48, 32, 6, 10
    base, cents, ix, assign, _, _ = _setup(oracle, ctx, n, d, nlist, nprobe, seed=77, kind="gauss")
    rng = np.random.default_rng(5)
    valid = rng.random(n) > 0.3
    filt = rng.random(n) < 0.2
    queries = rng.standard_normal((150, d)).astype(F)

    def lists_of(live):
        rows = np.flatnonzero(live)
        order = rows[np.argsort(assign[rows], kind="stable")].astype(np.uint64)
        offsets = np.zeros(nlist + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(np.bincount(assign[rows], minlength=nlist))
        return offsets, order

    h = ix._resident()
    vm = _mask_words(valid, n)
    _lib.check(_lib.lib().needle_hip_index_set_valid_mask(h, vm.ctypes.data))
    for precision in ("fp32", "bf16"):
        off, order = lists_of(valid)
        _check_search(oracle, ix, base, cents, off, order, queries, nprobe, k, precision=precision)
        off, order = lists_of(valid & filt)
        _check_search(oracle, ix, base, cents, off, order, queries, nprobe, k, precision=precision, filter_mask=filt)
    ids, _, counts = ix.batch_search_arrays(queries, k, filter_mask=np.zeros(n, dtype=bool))
    assert counts.tolist() == [0] * 150
    ix.close()


@pytest.mark.parametrize("shape", [(20000, 64, 32, 6, 8, 10), (8000, 96, 16, 16, 12, 5)])
def test_ivf_pq_lists(ctx, oracle, shape):
    """IvfConfig::use_pq: lists keep pq.encode(vector), the scan scores ProductQuantizer::asymmetric_distance
    (ivf.rs:488-491, 536-546).  Codes equal the oracle's, hits bit-equal, for every precision setting."""
    from needle_amd.ivf import IvfConfig, IvfIndex
    from needle_amd.quantization import ProductQuantizer

    n, d, nlist, nprobe, m, k = shape
    rng = np.random.default_rng(n + m)
    base = rng.standard_normal((n, d)).astype(F)
    cents = oracle.ivf_kmeans(base[: 20 * nlist], nlist, max_iter=5, seed=2)
    opq = oracle.ProductQuantizer.train(base[:2000], m, seed=4)
    codes = opq.encode(base)
    codes_dup = codes.copy()
    ix = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe).with_pq(m), ctx=ctx)
    ix.set_centroids(cents)
    ix.set_pq(ProductQuantizer(opq.codebooks))
    ix.add(range(n), base)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    queries = np.concatenate([base[:4], rng.standard_normal((90, d)).astype(F)])
    oi, od, oc = oracle.ivf_batch_search(queries, None, cents, offsets, order, nprobe, k, pq_codes=codes_dup,
                                         pq_codebooks=opq.codebooks, pq_m=m)
    for precision in ("fp32", "bf16", "exact"):
        ids, dists, counts = ix.batch_search_arrays(queries, k, precision=precision)
        assert np.array_equal(counts, oc), precision
        for i in range(queries.shape[0]):
            c = int(oc[i])
            assert np.array_equal(ids[i, :c], oi[i, :c]), (precision, i, ids[i, :c], oi[i, :c])
            assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (precision, i)
    # k beyond the candidate budget of the first pass: the exact scan over the codes
    oi, od, oc = oracle.ivf_batch_search(queries[:6], None, cents, offsets, order, nprobe, 300, pq_codes=codes_dup,
                                         pq_codebooks=opq.codebooks, pq_m=m)
    ids, dists, counts = ix.batch_search_arrays(queries[:6], 300)
    assert np.array_equal(counts, oc)
    for i in range(6):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]) and np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("case", ["duplicates", "near_ties", "nprobe_ge_nlist", "many_lists", "cluster40", "cluster200"])
def test_probe_set_decided_by_scores(ctx, oracle, case):
    """The search needs the probed lists as a SET: centroids whose first-pass score decides their membership are not
    re-scored, the ones near the nprobe-th are (flat_post.hip, set-only re-score).  Centroid sets that put many
    centroids at or next to the boundary -- exact duplicates, copies differing in the last bits, fewer lists than
    probes, and thousands of lists -- must give the oracle's hits all the same (ivf.rs:521-557)."""
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng({"duplicates": 5, "near_ties": 6, "nprobe_ge_nlist": 7, "many_lists": 8, "cluster40": 9,
                                 "cluster200": 10}[case])
    d = 64
    if case == "many_lists":
        n, nlist, nprobe = 40000, 1500, 24
        cents = rng.standard_normal((nlist, d)).astype(F)
    elif case == "nprobe_ge_nlist":
        n, nlist, nprobe = 4000, 12, 12
        cents = rng.standard_normal((nlist, d)).astype(F)
    elif case == "cluster40":
        # 40 centroids within rounding of each other and 56 far ones: more centroids between the two score margins than
        # the re-score stages at a time, with the rest of the candidates clearly out
        n, nlist, nprobe = 20000, 96, 8
        near = rng.standard_normal((1, d)).astype(F) * (F(1.0) + F(2e-6) * rng.standard_normal((40, d)).astype(F))
        cents = np.ascontiguousarray(np.concatenate([near, F(4.0) * rng.standard_normal((56, d)).astype(F)])[rng.permutation(nlist)])
    elif case == "cluster200":
        # 200 centroids within rounding of each other: more of them at the boundary than a query keeps candidates, so the
        # first-pass scores cannot settle the probe set.  The search has gone on behind the coarse step by then (it does not
        # wait for it): such queries are searched again behind an exact coarse step (metric: refined_queries).
        n, nlist, nprobe = 20000, 256, 8
        near = rng.standard_normal((1, d)).astype(F) * (F(1.0) + F(2e-6) * rng.standard_normal((200, d)).astype(F))
        cents = np.ascontiguousarray(np.concatenate([near, F(4.0) * rng.standard_normal((56, d)).astype(F)])[rng.permutation(nlist)])
    else:
        n, nlist, nprobe = 20000, 96, 8
        seeds = rng.standard_normal((12, d)).astype(F)
        cents = np.repeat(seeds, 8, axis=0)  # 12 groups of 8 identical centroids: every boundary falls inside a group
        if case == "near_ties":
            cents = cents * (F(1.0) + F(1e-6) * rng.standard_normal(cents.shape).astype(F))
        cents = np.ascontiguousarray(cents[rng.permutation(nlist)])
    base = (cents[rng.integers(0, nlist, n)] + F(0.3) * rng.standard_normal((n, d)).astype(F)).astype(F)
    ix = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(n), base)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    queries = np.concatenate([cents[:40] + F(0.05) * rng.standard_normal((min(40, nlist), d)).astype(F),
                              rng.standard_normal((200, d)).astype(F)]).astype(F)
    probes = ix.probe(queries)
    for i in range(0, queries.shape[0], 7):
        assert np.array_equal(probes[i], oracle.ivf_probe(queries[i], cents, nprobe)), i
    ctx.reset_metrics()
    for precision in ("bf16", "fp32"):
        _check_search(oracle, ix, base, cents, offsets, order, queries, nprobe, 10, precision=precision)
    if case == "cluster200":
        assert ctx.metrics()["refined_queries"] > 0
    ix.close()


def test_thresholds_come_from_rows_the_query_scans(ctx, oracle):
    """nprobe = 1 and two centroids a rounding apart: a query's starting threshold may only be taken from rows of the
    list it really probes.  The other list holds rows far nearer to the query -- a threshold sampled there (say, because a
    reduced-precision score ranked that centroid first) would rule out every row the query goes on to scan."""
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(77)
    d, eps = 64, F(1e-3)
    cents = np.zeros((2, d), dtype=F)
    cents[1, 0] = eps                                      # the centroids differ by 1e-3 in one coordinate
    near = np.zeros((300, d), dtype=F)                     # rows of list 0: five units from the queries
    near[:, 0] = F(-1e-3)
    near[:, 1] = F(5.0)
    near += F(1e-4) * rng.standard_normal(near.shape).astype(F)
    far = np.zeros((300, d), dtype=F)                      # rows of list 1: a hundred units away
    far[:, 0] = F(100.0)
    far += F(0.5) * rng.standard_normal(far.shape).astype(F)
    base = np.concatenate([near, far]).astype(F)
    queries = np.zeros((200, d), dtype=F)                  # on the line through the centroids, either side of the bisector:
    queries[:, 0] = eps / F(2.0) + F(2e-6) * rng.choice([-1.0, 1.0], 200).astype(F)   # f32 tells the sides apart, bf16 scores do not
    ix = IvfIndex(d, IvfConfig.new(2).with_nprobe(1), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(base.shape[0]), base)
    assign = oracle.ivf_assign(base, cents)
    assert np.all(assign[:300] == 0) and np.all(assign[300:] == 1)
    offsets, order = oracle.ivf_lists_from_assignment(assign, 2)
    probed = np.array([oracle.ivf_probe(q, cents, 1)[0] for q in queries])
    assert 40 < np.count_nonzero(probed == 1) < 160        # both sides occur
    for precision in ("bf16", "fp32"):
        _check_search(oracle, ix, base, cents, offsets, order, queries, 1, 5, precision=precision)
    ix.close()
