"""The benchmark configurations at their real parameters, through the C ABI:

  * the plumbing shape (100K x 128 Euclidean, 1K queries, k = 10) in full against the oracle;
  * IVF nlist = 4096, nprobe = 32, D = 384 over the 1.25M-row per-GPU share of the 10M-row configuration, in Euclidean
    and in unit-normalised ("cosine": IvfIndex ranks by Euclidean distance, src/indexing/ivf.rs:67, which over unit
    rows is the cosine order) form: identical probe lists and an oracle sample, plus the size-independent properties
    (idempotence, both list-scan precisions agree, two row shards merged == the whole index);
  * SQ8 and PQ (m = 48) over the 6.25M-row per-GPU share of the 50M-row configuration: an oracle sample (PQ: at full
    size; SQ8: over the first million rows) and the same properties;
  * cosine rows whose norm the reference's f32 arithmetic cannot represent (1e-23, 1e20), both first-pass precisions.

Ids bit-exact, distances bit-equal.
"""
import ctypes as C

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


def _search_handle(handle, dq, k, *, nprobe=0, precision="bf16"):
    import torch

    from needle_amd import _lib
    from needle_amd.index import PRECISION

    nq = dq.shape[0]
    ids = torch.empty((nq, k), dtype=torch.int64, device=dq.device)
    dists = torch.empty((nq, k), dtype=torch.float32, device=dq.device)
    counts = torch.empty((nq,), dtype=torch.int32, device=dq.device)
    sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0, filter_bitmask=None)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().needle_hip_batch_search_device(handle, dq.data_ptr(), nq, C.byref(sp), ids.data_ptr(),
                                                         dists.data_ptr(), counts.data_ptr()))
    return ids, dists, counts


def _same(a, b):
    import torch

    return all(torch.equal(x, y) for x, y in zip(a, b))


def _ordered_distinct(ids, dists, counts, k):
    """Every row: `counts` hits ascending by (distance, id), ids distinct."""
    assert bool((counts == k).all())
    d0, d1 = dists[:, :-1], dists[:, 1:]
    assert bool(((d0 < d1) | ((d0 == d1) & (ids[:, :-1] < ids[:, 1:]))).all())


def _merged_shards(ctx, handles, offsets, dq, k, **kw):
    import torch

    from needle_amd.sharded import hip_merge, pack_slot, slot_bytes

    q = dq.shape[0]
    slots = torch.zeros((len(handles), slot_bytes(q, k)), dtype=torch.uint8, device=dq.device)
    for w, (h, off) in enumerate(zip(handles, offsets)):
        ids, dists, counts = _search_handle(h, dq, k, **kw)
        pack_slot(slots[w], torch.where(ids >= 0, ids + off, ids), dists, counts)
    return hip_merge(ctx)(slots, len(handles), q, k)


def _bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint32)


def test_plumbing_shape_full_size(ctx, oracle):
    """100K x 128 Euclidean, 1K queries, k = 10: every query against the oracle, both first-pass precisions."""
    import needle_amd as na

    rng = np.random.default_rng(1)
    base = rng.standard_normal((100_000, 128)).astype(F)
    queries = rng.standard_normal((1000, 128)).astype(F)
    oi, od, oc = oracle.flat_batch_search(queries, base, 10, oracle.EUCLIDEAN, threads=min(oracle.cpu_threads(), 16))
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    for precision in ("bf16", "fp32"):
        ids, dists, counts = ix.search(queries, 10, precision=precision)
        assert np.array_equal(counts, oc) and np.array_equal(ids, oi), precision
        assert np.array_equal(_bits(dists), _bits(od)), precision
    ix.close()


@pytest.mark.parametrize("form", ["euclidean", "unit-normalised"])
def test_ivf_configuration_parameters(ctx, oracle, form):
    import torch

    from needle_amd import _lib
    from needle_amd.ivf import IvfConfig, IvfIndex

    n, dim, nlist, nprobe, nq, k = 1_250_000, 384, 4096, 32, 2000, 10
    rng = np.random.default_rng(33)
    base = rng.standard_normal((n, dim), dtype=F)
    queries = rng.standard_normal((nq, dim), dtype=F)
    if form == "unit-normalised":
        base, queries = ctx.normalize(base), ctx.normalize(queries)
    # centroids: IvfIndex::train on the device over 100 rows per list (bit-equal to the seeded restatement, test_ivf_gpu.py)
    trainer = IvfIndex(dim, IvfConfig.new(nlist).with_max_iterations(4), ctx=ctx, seed=3)
    trainer.train(base[rng.choice(n, 100 * nlist, replace=False)])
    cents = trainer.centroids
    trainer.close()
    L = _lib.lib()

    def build(rows):
        h = C.c_void_p()
        _lib.check(L.needle_hip_ivf_build(ctx._h, rows.ctypes.data, rows.shape[0], dim, cents.ctypes.data, nlist, None, None,
                                          C.byref(h)))
        return h

    whole = build(base)
    dq = torch.from_numpy(queries).to(torch.device("cuda", 0))
    first = _search_handle(whole, dq, k, nprobe=nprobe, precision="bf16")
    _ordered_distinct(*first, k)
    assert _same(first, _search_handle(whole, dq, k, nprobe=nprobe, precision="bf16"))  # idempotent
    assert _same(first, _search_handle(whole, dq, k, nprobe=nprobe, precision="fp32"))  # the other list-scan precision
    # identical probe lists, and the oracle's hits on a sample (its lists built from the device assignment, which
    # test_ivf_gpu.py checks bit for bit against the oracle's own)
    s = 24
    probe = np.full((s, nprobe), 0xFFFFFFFF, dtype=np.uint32)
    pc = np.zeros(s, dtype=np.uint32)
    _lib.check(L.needle_hip_ivf_probe(whole, queries.ctypes.data, s, dim, nprobe, probe.ctypes.data, pc.ctypes.data))
    want_probe = np.stack([oracle.ivf_probe(queries[i], cents, nprobe) for i in range(s)])
    assert np.array_equal(probe.astype(np.int64), want_probe.astype(np.int64))
    assign = ctx.kmeans_assign(base, cents, 0)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    oi, od, oc = oracle.ivf_batch_search(queries[:s], base, cents, offsets, order, nprobe, k)
    assert np.array_equal(first[0][:s].cpu().numpy().view(np.uint64), oi)
    assert np.array_equal(_bits(first[1][:s].cpu().numpy()), _bits(od))
    assert np.array_equal(first[2][:s].cpu().numpy().view(np.uint32), oc)
    # recall@10 against the exact flat answer equals the oracle's IVF recall by construction (same ids); it is high
    L.needle_hip_index_destroy(whole)
    # two row shards with replicated centroids, merged: the whole index's result
    half = n // 2
    shards = [build(np.ascontiguousarray(base[:half])), build(np.ascontiguousarray(base[half:]))]
    merged = _merged_shards(ctx, shards, [0, half], dq, k, nprobe=nprobe, precision="bf16")
    assert _same(first, merged)
    for h in shards:
        L.needle_hip_index_destroy(h)


def test_sq8_per_gpu_share(ctx, oracle):
    import torch

    from needle_amd.quantization import ScalarQuantizer, Sq8Index

    n, dim, nq, k = 6_250_000, 384, 1000, 10
    rng = np.random.default_rng(13)
    sample = rng.standard_normal((20000, dim), dtype=F)
    sq = ScalarQuantizer.train(sample)
    codes = rng.integers(0, 256, size=(n, dim), dtype=np.uint8)
    codes[:20000] = sq.quantize(sample)
    queries = rng.standard_normal((nq, dim), dtype=F)
    dq = torch.from_numpy(queries).to(torch.device("cuda", 0))
    ix = Sq8Index(ctx, sq, codes)
    first = _search_handle(ix._h, dq, k, precision="bf16")
    _ordered_distinct(*first, k)
    assert _same(first, _search_handle(ix._h, dq, k, precision="bf16"))
    assert _same((first[0][:200], first[1][:200], first[2][:200]), _search_handle(ix._h, dq[:200], k, precision="fp32"))
    ix.close()
    # the oracle's sample: its scalar decode-and-divide loop costs seconds per query and million rows, so it scans the
    # first million code rows (an index of their own) for 8 queries; the full size is held by the properties
    s, ns = 8, 1_000_000
    sub = Sq8Index(ctx, sq, codes[:ns])
    gi, gd, gc = _search_handle(sub._h, dq[:s], k, precision="bf16")
    osq = oracle.ScalarQuantizer(sq.min_vals, sq.max_vals, sq.scale, 0.0)
    oi, od, oc = osq.batch_search(queries[:s], codes[:ns], k, threads=min(oracle.cpu_threads(), 16))
    assert np.array_equal(gi.cpu().numpy().view(np.uint64), oi)
    assert np.array_equal(_bits(gd.cpu().numpy()), _bits(od))
    sub.close()
    half = n // 2
    shards = [Sq8Index(ctx, sq, codes[:half]), Sq8Index(ctx, sq, codes[half:])]
    merged = _merged_shards(ctx, [sh._h for sh in shards], [0, half], dq, k, precision="bf16")
    assert _same(first, merged)
    for sh in shards:
        sh.close()


def test_pq_per_gpu_share(ctx, oracle):
    import torch

    from needle_amd.quantization import PqIndex, ProductQuantizer

    n, dim, m, nq, k = 6_250_000, 384, 48, 1000, 10
    rng = np.random.default_rng(17)
    books = rng.standard_normal((m, 256, dim // m), dtype=F)
    pq = ProductQuantizer(books)
    codes = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    queries = rng.standard_normal((nq, dim), dtype=F)
    dq = torch.from_numpy(queries).to(torch.device("cuda", 0))
    ix = PqIndex(ctx, pq, codes)
    first = _search_handle(ix._h, dq, k, precision="bf16")
    _ordered_distinct(*first, k)
    assert _same(first, _search_handle(ix._h, dq, k, precision="bf16"))
    assert _same((first[0][:100], first[1][:100], first[2][:100]), _search_handle(ix._h, dq[:100], k, precision="fp32"))
    s = 8
    opq = oracle.ProductQuantizer(books, np.full(m, 256, dtype=np.uint32), dim, m)
    oi, od, oc = opq.batch_search(queries[:s], codes, k, threads=min(oracle.cpu_threads(), 16))
    assert np.array_equal(first[0][:s].cpu().numpy().view(np.uint64), oi)
    assert np.array_equal(_bits(first[1][:s].cpu().numpy()), _bits(od))
    ix.close()
    half = n // 2
    shards = [PqIndex(ctx, pq, codes[:half]), PqIndex(ctx, pq, codes[half:])]
    merged = _merged_shards(ctx, [sh._h for sh in shards], [0, half], dq, k, precision="bf16")
    assert _same(first, merged)
    for sh in shards:
        sh.close()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_cosine_rows_with_unrepresentable_norms(ctx, oracle, precision):
    """Rows scaled by 1e-23 (the reference's f32 dot(b, b) underflows: norm 0, distance forced to 1.0) and by 1e20 (it
    overflows: norm inf, distance 1 - dot / inf): no bound ties a first-pass score to such a distance, so these rows stay
    out of the first pass and every query re-scores them beside its own candidates -- the result is the reference
    arithmetic's, and no query falls back to the exact scan.  Small base (dense score matrix) and large (candidate lists);
    with masks; after an append that brings more such rows; and with more of them than a search can carry (exact scan)."""
    import needle_amd as na

    rng = np.random.default_rng(91)
    for n in (6000, 30000):
        base = rng.standard_normal((n, 64)).astype(F)
        queries = rng.standard_normal((40, 64)).astype(F)
        queries[:10] = base[100:110] + F(0.01) * rng.standard_normal((10, 64)).astype(F)  # nearest rows are about to be rescaled
        base[100:105] *= F(1e-23)
        base[105:110] *= F(1e20)
        base[n - 1] *= F(1e-23)
        valid = np.ones(n, dtype=bool)
        valid[[101, 106, 500]] = False
        filt = rng.random(n) < 0.7
        filt[100:110:2] = True
        ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine, valid=valid)
        ctx.reset_metrics()
        for fm in (None, filt):
            oi, od, oc = oracle.flat_batch_search(queries, base, 10, oracle.COSINE, valid=valid, filt=fm)
            ids, dists, counts = ix.search(queries, 10, precision=precision, filter_mask=fm)
            assert np.array_equal(counts, oc) and np.array_equal(ids, oi), (n, fm is not None)
            assert np.array_equal(_bits(dists), _bits(od))
        assert ctx.metrics()["fallback_queries"] == 0
        # an append with two more such rows: the side list follows
        more = rng.standard_normal((300, 64)).astype(F)
        more[7] *= F(1e21)
        more[8] = queries[3] * F(1e-24)
        ix.append(more)
        all_rows = np.concatenate([base, more])
        all_valid = np.concatenate([valid, np.ones(300, dtype=bool)])
        oi, od, oc = oracle.flat_batch_search(queries, all_rows, 10, oracle.COSINE, valid=all_valid)
        ids, dists, counts = ix.search(queries, 10, precision=precision)
        assert np.array_equal(counts, oc) and np.array_equal(ids, oi) and np.array_equal(_bits(dists), _bits(od))
        assert ctx.metrics()["fallback_queries"] == 0
        ix.close()
    # 250 such rows are more than a search carries as extra candidates: the exact scan answers, still the reference's bits
    base = rng.standard_normal((5000, 32)).astype(F)
    base[1000:1250] *= F(1e-22)
    queries = rng.standard_normal((12, 32)).astype(F)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    oi, od, oc = oracle.flat_batch_search(queries, base, 10, oracle.COSINE)
    ids, dists, counts = ix.search(queries, 10, precision=precision)
    assert np.array_equal(counts, oc) and np.array_equal(ids, oi) and np.array_equal(_bits(dists), _bits(od))
    ix.close()
