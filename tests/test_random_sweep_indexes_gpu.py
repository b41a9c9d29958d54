"""Seeded random sweep of the IVF, IVF-PQ, SQ8, PQ and binary indexes through the C ABI: shape, k, probe count,
first-pass precision, row masks and the kind of data drawn together, so that list regrouping, mask folding, candidate
pruning, the refinement chain and the tie rules meet in combinations the targeted tests do not spell out.  Same bar as
everywhere: ids bit-exact, distances bit-equal to the oracle, unused slots padded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# NEEDLE_TEST_SEED_OFFSET shifts every generator seed: the same sweep over fresh cases (fuzzing runs)
SEED_OFFSET = int(__import__("os").environ.get("NEEDLE_TEST_SEED_OFFSET", "0"))

F = np.float32
U64MAX = np.iinfo(np.uint64).max


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _rows(rng, n, dim, kind):
    if kind == "uniform":
        return rng.random((n, dim), dtype=F)
    x = rng.standard_normal((n, dim)).astype(F)
    if kind == "clustered":  # a few tight blobs, rows in blob order
        centres = rng.standard_normal((5, dim)).astype(F) * F(3.0)
        x = (centres[np.sort(rng.integers(0, 5, n))] + x * F(0.05)).astype(F)
    if kind == "scaled":
        x = (x * F(rng.choice([1e-2, 40.0]))).astype(F)
    return x


def _masks(rng, n):
    valid = (rng.random(n) > rng.choice([0.1, 0.5])) if rng.random() < 0.5 else None
    filt = (rng.random(n) < rng.choice([0.02, 0.5, 0.9])) if rng.random() < 0.5 else None
    return valid, filt


def _live(n, valid, filt):
    live = np.ones(n, dtype=bool)
    if valid is not None:
        live &= valid
    if filt is not None:
        live &= filt
    return live


def _cmp(got, want, tag):
    ids, dists, counts = got
    oi, od, oc = want
    assert np.array_equal(counts, oc), (tag, counts[:8], oc[:8])
    for i in range(ids.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (tag, i, ids[i, :c][:6], oi[i, :c][:6])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (tag, i)
        assert np.all(ids[i, c:] == U64MAX), (tag, i)


def _set_valid(ix_handle, valid, n):
    from needle_amd import _lib
    from needle_amd.index import _mask_words

    vm = _mask_words(valid, n)
    _lib.check(_lib.lib().needle_hip_index_set_valid_mask(ix_handle, vm.ctypes.data))


@pytest.mark.parametrize("seed", range(16))
def test_random_ivf_case(ctx, oracle, seed):
    """IVF-Flat and IVF-PQ lists: the oracle scans, in list order, the live rows of the probed lists."""
    from needle_amd.ivf import IvfConfig, IvfIndex
    from needle_amd.quantization import ProductQuantizer

    rng = np.random.default_rng(5000 + seed + SEED_OFFSET)
    use_pq = rng.random() < 0.35
    dim = int(rng.choice([8, 24, 48, 96, 128, 384] if use_pq else [3, 17, 48, 100, 128, 384]))
    n = int(rng.choice([40, 900, 6000, 30000, 90000]))
    nlist = int(min(n, rng.choice([1, 5, 16, 64, 200])))
    nprobe = int(min(nlist, rng.choice([1, 2, 6, 20, 200])))
    nq = int(rng.choice([1, 7, 70, 300]))
    k = int(rng.choice([1, 5, 10, 50, 200]))
    kind = str(rng.choice(["gauss", "uniform", "clustered", "scaled"]))
    precision = str(rng.choice(["fp32", "bf16", "bf16", "exact"]))
    base, queries = _rows(rng, n, dim, kind), _rows(rng, nq, dim, kind)
    if rng.random() < 0.3 and n > 20:  # exact duplicates: ties decided by list position
        base[n // 2: n // 2 + 9] = base[3]
        queries[0] = base[3]
    cents = oracle.ivf_kmeans(base[: min(n, 20 * nlist)], nlist, max_iter=4, seed=seed)
    nlist = cents.shape[0]
    nprobe = min(nprobe, nlist)
    cfg = IvfConfig.new(nlist).with_nprobe(nprobe)
    m = 0
    if use_pq:
        m = int(rng.choice([d for d in (2, 4, 8, 12) if dim % d == 0]))
        opq = oracle.ProductQuantizer.train(base[: min(n, 1500)], m, seed=seed)
        codes = opq.encode(base)
        cfg = cfg.with_pq(m)
    ix = IvfIndex(dim, cfg, ctx=ctx)
    ix.set_centroids(cents)
    if use_pq:
        ix.set_pq(ProductQuantizer(opq.codebooks))
    ix.add(range(n), base)
    assign = oracle.ivf_assign(base, cents)
    valid, filt = _masks(rng, n)
    if valid is not None:
        _set_valid(ix._resident(), valid, n)
    rows = np.flatnonzero(_live(n, valid, filt))
    order = rows[np.argsort(assign[rows], kind="stable")].astype(np.uint64)
    offsets = np.zeros(nlist + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(np.bincount(assign[rows], minlength=nlist))
    if use_pq:
        want = oracle.ivf_batch_search(queries, None, cents, offsets, order, nprobe, k, pq_codes=codes.copy(),
                                       pq_codebooks=opq.codebooks, pq_m=m)
    else:
        want = oracle.ivf_batch_search(queries, base, cents, offsets, order, nprobe, k)
    tag = ("ivf", use_pq, m, dim, n, nlist, nprobe, nq, k, kind, precision, valid is not None, filt is not None)
    try:
        for _ in range(2):
            _cmp(ix.batch_search_arrays(queries, k, precision=precision, filter_mask=filt), want, tag)
    finally:
        ix.close()


@pytest.mark.parametrize("seed", range(18))
def test_random_code_scan_case(ctx, oracle, seed):
    """SQ8, PQ and binary code scans."""
    from needle_amd.quantization import (BinaryIndex, BinaryQuantizer, PqIndex, ProductQuantizer, ScalarQuantizer,
                                         Sq8Index)

    rng = np.random.default_rng(7000 + seed + SEED_OFFSET)
    which = ["sq8", "pq", "binary"][seed % 3]
    n = int(rng.choice([3, 200, 5000, 40000, 120000]))
    nq = int(rng.choice([1, 9, 64, 200]))
    k = int(rng.choice([1, 5, 10, 100, 300, 700]))
    kind = str(rng.choice(["gauss", "uniform", "clustered", "scaled"]))
    if which == "pq":
        dim = int(rng.choice([8, 24, 48, 96, 128, 384]))
    else:
        dim = int(rng.choice([5, 24, 64, 100, 384, 700]))
    while float(n) * dim * nq > 3e9:
        nq = max(1, nq // 2)
    base, queries = _rows(rng, n, dim, kind), _rows(rng, nq, dim, kind)
    valid, filt = _masks(rng, n)
    live = _live(n, valid, None)
    flive = _live(n, valid, filt)
    dup = rng.random() < 0.4 and n > 300
    train = base[: max(min(n, 50), n // 10)]
    if which == "sq8":
        o = oracle.ScalarQuantizer.train(train)
        codes = o.quantize(base)
        if dup:
            codes[n // 2: n // 2 + 150] = codes[2]
            queries[0] = base[2]
        precision = str(rng.choice(["fp32", "bf16", "bf16", "exact"]))
        ix = Sq8Index(ctx, ScalarQuantizer(o.min_vals, o.max_vals, o.scale), codes, valid=valid)
    elif which == "pq":
        m = int(rng.choice([d for d in (2, 4, 8, 12, 48) if dim % d == 0]))
        o = oracle.ProductQuantizer.train(base[: min(n, 1200)], m, seed=seed)
        codes = o.encode(base)
        if dup:
            codes[n // 2: n // 2 + 150] = codes[2]
            queries[0] = base[2]
        precision = str(rng.choice(["fp32", "bf16", "bf16"]))
        ix = PqIndex(ctx, ProductQuantizer(o.codebooks), codes, valid=valid)
    else:
        o = oracle.BinaryQuantizer.train(train)
        codes = o.quantize(base)
        if dup:
            codes[n // 2: n // 2 + 150] = codes[2]
            queries[0] = base[2]
        precision = None
        ix = BinaryIndex(ctx, BinaryQuantizer(o.thresholds), codes, valid=valid)
    tag = (which, dim, n, nq, k, kind, precision, valid is not None, filt is not None, dup)
    kw = {} if precision is None else {"precision": precision}
    try:
        want = o.batch_search(queries, codes, k, valid=None if valid is None else live)
        for _ in range(2):
            _cmp(ix.search(queries, k, **kw), want, tag)
        if filt is not None:
            want = o.batch_search(queries, codes, k, valid=flive)
            _cmp(ix.search(queries, k, filter_mask=filt, **kw), want, tag + ("filtered",))
    finally:
        ix.close()
