"""Seeded random sweep of the flat search through the C ABI: metric, dimension, row count, k, first-pass precision
and row masks drawn together, so that the pieces that share the path (threshold seeding, staged launches, masks folded
into the accumulator seeds, candidate pruning, adaptive candidate slack, the refinement chain) meet in combinations
the targeted tests do not spell out.  Same bar as everywhere: ids bit-exact, distances bit-equal to the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# NEEDLE_TEST_SEED_OFFSET shifts every generator seed: the same sweep over fresh cases (fuzzing runs)
SEED_OFFSET = int(__import__("os").environ.get("NEEDLE_TEST_SEED_OFFSET", "0"))

F = np.float32
U64MAX = np.iinfo(np.uint64).max
METRICS = ["Cosine", "CosineNormalized", "Euclidean", "DotProduct", "Manhattan", "Chebyshev", "Hamming"]


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _case(rng):
    # four in five cases on the matrix-core metrics, where seeding, staging, mask folding and pruning live
    metric = str(rng.choice(METRICS[:4])) if rng.random() < 0.8 else str(rng.choice(METRICS[4:]))
    dim = int(rng.choice([3, 17, 32, 64, 100, 128, 200, 384, 500, 768, 900]))
    n = int(rng.choice([7, 130, 5000, 20000, 70000, 200000]))
    nq = int(rng.choice([1, 2, 33, 130, 300]))
    if metric in ("Manhattan", "Chebyshev", "Hamming"):  # exact scan only: keep the oracle and the scan short
        n, dim = min(n, 5000), min(dim, 200)
    while float(n) * dim * nq > 6e9:  # bounds the oracle's scalar work per call
        nq = max(1, nq // 2) if nq > 33 else nq
        dim = dim if nq > 33 else max(32, dim // 2)
    k = int(rng.choice([1, 5, 10, 64, 100, 250, 400]))
    precision = str(rng.choice(["fp32", "bf16", "bf16", "exact"])) if n <= 20000 else str(rng.choice(["fp32", "bf16", "bf16"]))
    kind = str(rng.choice(["gauss", "uniform", "clustered", "scaled"]))
    # few slices make them long: pre-pass thresholds and staged launches (a development switch of the library)
    splits = int(rng.choice([0, 0, 4, 8])) if n >= 70000 else 0
    return metric, dim, n, nq, k, precision, kind, splits


def _rows(rng, n, dim, kind):
    if kind == "uniform":
        return rng.random((n, dim), dtype=F)
    x = rng.standard_normal((n, dim)).astype(F)
    if kind == "clustered":  # a few tight blobs, rows in blob order: close scores, clustered arrival
        centres = rng.standard_normal((5, dim)).astype(F) * F(3.0)
        x = (centres[np.sort(rng.integers(0, 5, n))] + x * F(0.05)).astype(F)
    if kind == "scaled":     # magnitudes far from 1
        x = (x * F(rng.choice([1e-3, 40.0]))).astype(F)
    return x


@pytest.mark.parametrize("seed", range(36))
def test_random_flat_case(ctx, oracle, seed):
    import needle_amd as na

    rng = np.random.default_rng(1000 + seed + SEED_OFFSET)
    metric, dim, n, nq, k, precision, kind, splits = _case(rng)
    base, queries = _rows(rng, n, dim, kind), _rows(rng, nq, dim, kind)
    if metric == "CosineNormalized":
        base, queries = oracle.normalize_rows(base), oracle.normalize_rows(queries)
    if rng.random() < 0.3 and n > 20:  # exact duplicates: ties decided by row id
        base[n // 2: n // 2 + 9] = base[3]
        queries[0] = base[3]
    valid = (rng.random(n) > 0.3) if rng.random() < 0.5 else None
    filt = (rng.random(n) < rng.choice([0.02, 0.5, 0.9])) if rng.random() < 0.5 else None
    m = na.DistanceFunction[metric]
    ix = na.FlatIndex(ctx, base, m, valid=valid)
    tag = (metric, dim, n, nq, k, precision, kind, splits, valid is not None, filt is not None)
    if splits:
        ctx.set_tuning(splits=splits)
    if filt is not None and rng.random() < 0.6:  # small bases too take the gathered copy of the admitted rows
        ctx.set_tuning(compact_min_rows=1, compact_pct=int(rng.choice([50, 95])))
    try:
        _run_twice(oracle, ix, queries, base, k, m, precision, valid, filt, nq, tag)
    finally:
        ctx.set_tuning(splits=0, compact_min_rows=0, compact_pct=0)
        ix.close()


def _run_twice(oracle, ix, queries, base, k, m, precision, valid, filt, nq, tag):
    oi, od, oc = oracle.flat_batch_search(queries, base, k, int(m), valid=valid, filt=filt)
    for _ in range(2):  # twice: the second search runs with whatever the first one adapted
        ids, dists, counts = ix.search(queries, k, precision=precision, filter_mask=filt)
        assert np.array_equal(counts, oc), tag
        for i in range(nq):
            c = int(oc[i])
            assert np.array_equal(ids[i, :c], oi[i, :c]), (tag, i, ids[i, :c][:6], oi[i, :c][:6])
            assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (tag, i)
            assert np.all(ids[i, c:] == U64MAX), (tag, i)


@pytest.mark.parametrize("seed", range(10))
def test_random_resident_index_history(ctx, oracle, seed):
    """A resident flat index lives through a random history -- appends of any size (host or device rows, with or without
    masks), swap-removes, filtered and unfiltered searches in either precision -- and answers every search like the
    oracle over the rows it holds at that moment."""
    import torch

    import needle_amd as na

    rng = np.random.default_rng(9000 + seed + SEED_OFFSET)
    metric = str(rng.choice(METRICS[:4]))
    dim = int(rng.choice([5, 32, 96, 200]))
    m = na.DistanceFunction[metric]

    def draw(count):
        x = rng.standard_normal((count, dim)).astype(F)
        if metric == "CosineNormalized":
            x = oracle.normalize_rows(x)
        return x

    rows = draw(int(rng.choice([0, 1, 300, 5000])))
    valid = np.ones(rows.shape[0], dtype=bool)
    ix = na.FlatIndex(ctx, rows, m, dim=dim)
    try:
        for step in range(int(rng.integers(6, 12))):
            op = rng.random()
            if op < 0.45 or rows.shape[0] < 20:
                add = draw(int(rng.choice([1, 7, 130, 2000, 9000])))
                if metric == "Cosine" and rng.random() < 0.3:
                    add[rng.integers(0, add.shape[0])] *= F(1e-23)   # a norm f32 cannot square
                mask = (rng.random(add.shape[0]) > 0.2) if rng.random() < 0.4 else None
                src = torch.from_numpy(add).to("cuda:0") if rng.random() < 0.3 else add
                ix.append(src, valid=mask)
                rows = np.concatenate([rows, add])
                valid = np.concatenate([valid, mask if mask is not None else np.ones(add.shape[0], dtype=bool)])
            elif op < 0.6:
                pos = int(rng.integers(0, rows.shape[0]))
                ix.swap_remove(pos)
                rows[pos], valid[pos] = rows[-1], valid[-1]
                rows, valid = rows[:-1].copy(), valid[:-1].copy()
            n = rows.shape[0]
            assert len(ix) == n
            nq, k = int(rng.choice([1, 9, 70])), int(rng.choice([1, 10, 60]))
            queries = draw(nq)
            if n > 3 and rng.random() < 0.5:
                queries[0] = rows[int(rng.integers(0, n))]
            filt = (rng.random(n) < rng.choice([0.03, 0.4, 0.9])) if rng.random() < 0.6 else None
            if filt is not None and rng.random() < 0.7:
                ctx.set_tuning(compact_min_rows=1)
            precision = str(rng.choice(["bf16", "fp32"]))
            tag = (seed, step, metric, dim, n, nq, k, precision, filt is not None)
            oi, od, oc = oracle.flat_batch_search(queries, rows, k, int(m), valid=valid, filt=filt)
            ids, dists, counts = ix.search(queries, k, precision=precision, filter_mask=filt)
            ctx.set_tuning(compact_min_rows=0)
            assert np.array_equal(counts, oc), tag
            for i in range(nq):
                c = int(oc[i])
                assert np.array_equal(ids[i, :c], oi[i, :c]), (tag, i, ids[i, :c][:6], oi[i, :c][:6])
                assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), (tag, i)
    finally:
        ctx.set_tuning(compact_min_rows=0)
        ix.close()
