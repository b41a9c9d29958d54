"""Incremental insert into a resident flat index (needle_hip_index_append, needle_hip_index_swap_remove; the
reference's GpuResidentIndex::{add, add_batch, remove}, src/experimental/gpu/mod.rs:1102-1207): an index built from N
rows and grown by appends answers every search exactly as an index built from all rows at once -- and as the oracle."""
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


def _same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and
            np.array_equal(a[2], b[2]))


def _oracle_equal(oracle, got, base, queries, k, metric, valid=None):
    oi, od, oc = oracle.flat_batch_search(queries, base, k, int(metric), valid=valid)
    ids, dists, counts = got
    assert np.array_equal(counts, oc)
    for i in range(queries.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), i


@pytest.mark.parametrize("metric", ["Cosine", "Euclidean", "DotProduct"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_build_then_append_equals_one_build(ctx, oracle, metric, precision):
    import needle_amd as na

    rng = np.random.default_rng(41)
    n0, m1, m2, d, k = 2500, 1300, 777, 96, 10
    rows = rng.standard_normal((n0 + m1 + m2, d)).astype(F)
    rows[n0 + 5] = rows[17]            # a duplicate across the append boundary: tie broken by internal id
    rows[n0 + m1 + 3] = rows[17]
    queries = rng.standard_normal((150, d)).astype(F)
    queries[0] = rows[17]
    m = na.DistanceFunction[metric]
    valid = rng.random(rows.shape[0]) > 0.2   # deleted rows in every part
    grown = na.FlatIndex(ctx, rows[:n0], m, valid=valid[:n0])
    before = grown.search(queries, k, precision=precision)  # (the bf16 operand copy exists before the appends)
    _oracle_equal(oracle, before, rows[:n0], queries, k, m, valid=valid[:n0])
    grown.append(rows[n0:n0 + m1], valid=valid[n0:n0 + m1])
    grown.append(rows[n0 + m1:], valid=valid[n0 + m1:])
    assert len(grown) == rows.shape[0]
    whole = na.FlatIndex(ctx, rows, m, valid=valid)
    for prec in (precision, "fp32" if precision == "bf16" else "bf16"):
        a, b = grown.search(queries, k, precision=prec), whole.search(queries, k, precision=prec)
        assert _same(a, b), prec
        _oracle_equal(oracle, a, rows, queries, k, m, valid=valid)
    grown.close()
    whole.close()


def test_many_small_appends_and_an_empty_start(ctx, oracle):
    """Capacity doubles several times; the index starts with no rows and no mask, a mask arrives with a later batch."""
    import torch

    import needle_amd as na

    rng = np.random.default_rng(42)
    d, k = 40, 7
    rows = rng.standard_normal((1900, d)).astype(F)
    queries = rng.standard_normal((33, d)).astype(F)
    ix = na.FlatIndex(ctx, np.zeros((0, d), dtype=F), na.DistanceFunction.Euclidean, dim=d)
    got = ix.search(queries, k)
    assert np.all(got[2] == 0)
    valid = np.ones(rows.shape[0], dtype=bool)
    at = 0
    for step, m in enumerate([1, 50, 77, 128, 300, 44, 1300]):
        part = rows[at:at + m]
        if step == 3:                                   # the first mask: everything before it stays live
            valid[at:at + m:3] = False
            ix.append(part, valid=valid[at:at + m])
        elif step == 5:                                 # device-resident rows
            ix.append(torch.from_numpy(part).to("cuda:0"))
        else:
            ix.append(part)
        at += m
        for prec in ("bf16", "fp32"):
            _oracle_equal(oracle, ix.search(queries, k, precision=prec), rows[:at], queries, k, na.DistanceFunction.Euclidean,
                          valid=valid[:at])
    assert at == rows.shape[0] and len(ix) == at
    ix.close()


def test_rejected_batch_leaves_the_index_as_it_was(ctx, oracle):
    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(43)
    d, k = 32, 5
    rows = rng.standard_normal((300, d)).astype(F)
    queries = rng.standard_normal((20, d)).astype(F)
    ix = na.FlatIndex(ctx, rows[:200], na.DistanceFunction.Cosine)
    bad = rows[200:260].copy()
    bad[7, 3] = np.nan
    with pytest.raises(_lib.NeedleHipError) as e:
        ix.append(bad)
    assert "Invalid vector" in str(e.value) and len(ix) == 200
    _oracle_equal(oracle, ix.search(queries, k, precision="bf16"), rows[:200], queries, k, na.DistanceFunction.Cosine)
    ix.append(rows[200:])   # the rows behind the index are clean again
    _oracle_equal(oracle, ix.search(queries, k, precision="bf16"), rows, queries, k, na.DistanceFunction.Cosine)
    _oracle_equal(oracle, ix.search(queries, k, precision="fp32"), rows, queries, k, na.DistanceFunction.Cosine)
    with pytest.raises(_lib.DimensionMismatch):
        ix.append(rows[:3, :5])
    ix.close()


def test_swap_remove_is_vec_swap_remove(ctx, oracle):
    import needle_amd as na

    rng = np.random.default_rng(44)
    d, k = 48, 6
    rows = [r for r in rng.standard_normal((700, d)).astype(F)]
    rows[650] = rows[3].copy()   # duplicates: their order after a removal decides the tie
    rows[699] = rows[3].copy()
    queries = rng.standard_normal((25, d)).astype(F)
    queries[0] = rows[3]
    valid = [True] * 700
    valid[10] = False
    valid[698] = False
    ix = na.FlatIndex(ctx, np.stack(rows), na.DistanceFunction.Euclidean, valid=np.array(valid))
    ix.search(queries, k, precision="bf16")
    for pos in (3, 0, 128, 511, len(rows) - 4):
        pos = min(pos, len(rows) - 1)
        rows[pos] = rows[-1]
        rows.pop()
        valid[pos] = valid[-1]
        valid.pop()
        ix.swap_remove(pos)
        assert len(ix) == len(rows)
        for prec in ("bf16", "fp32"):
            _oracle_equal(oracle, ix.search(queries, k, precision=prec), np.stack(rows), queries, k, na.DistanceFunction.Euclidean,
                          valid=np.array(valid))
    ix.append(np.stack(rows[:140]))  # grows again after shrinking across a tile boundary
    _oracle_equal(oracle, ix.search(queries, k, precision="bf16"), np.stack(rows + rows[:140]), queries, k,
                  na.DistanceFunction.Euclidean, valid=np.array(valid + [True] * 140))
    ix.close()


def test_resident_index_follows_the_reference_under_ties(ctx, oracle):
    """GpuResidentIndex: add_batch -> search -> add -> remove (swap_remove) -> search, with duplicated vectors, against
    the reference's own steps on the host (positions, then a stable sort by distance)."""
    from needle_amd.accelerator import GpuResidentIndex
    from needle_amd.distance import DistanceType

    rng = np.random.default_rng(45)
    d = 24
    vecs = rng.standard_normal((260, d)).astype(F)
    ix = GpuResidentIndex(d, DistanceType.Euclidean, ctx=ctx)
    ids, rows = [], []

    def add(i, v):
        ix.add(i, v)
        ids.append(i)
        rows.append(v)

    def remove(i):
        assert ix.remove(i)
        p = ids.index(i)
        ids[p], rows[p] = ids[-1], rows[-1]
        ids.pop()
        rows.pop()

    def check(q, k):
        si, sd = oracle.accel_fused_search(q, np.stack(rows), k, oracle.ACCEL_EUCLIDEAN)  # (distance, position) order
        got = ix.search(q, k)
        assert [h[0] for h in got] == [ids[int(j)] for j in si]
        assert np.array_equal(np.array([h[1] for h in got], dtype=F).view(np.uint32), sd.view(np.uint32))

    for i in range(200):
        add(1000 + i, vecs[i])
    check(vecs[5], 4)
    for i in range(200, 260):
        add(1000 + i, vecs[5] if i % 20 == 0 else vecs[i])   # duplicates of row 5 behind the resident rows
    check(vecs[5], 5)
    remove(1005)      # the last vector (no duplicate) takes position 5
    check(vecs[5], 5)
    remove(1220)      # a duplicate goes, the last one moves in
    add(5000, vecs[5])
    check(vecs[5], 6)
    assert not ix.remove(42) and len(ix) == len(ids)


def test_append_is_for_flat_indexes(ctx, oracle):
    import ctypes as C

    from needle_amd import _lib
    from needle_amd.ivf import IvfConfig, IvfIndex

    rng = np.random.default_rng(46)
    base = rng.standard_normal((600, 16)).astype(F)
    ix = IvfIndex(16, IvfConfig.new(8).with_nprobe(2), ctx=ctx)
    ix.set_centroids(oracle.ivf_kmeans(base[:200], 8, max_iter=3, seed=1))
    ix.add(range(600), base)
    h = ix._resident()
    st = _lib.lib().needle_hip_index_append(h, base.ctypes.data, 10, None)
    assert st == _lib.ERR_UNSUPPORTED
    assert _lib.lib().needle_hip_index_swap_remove(h, C.c_uint64(1)) == _lib.ERR_UNSUPPORTED
    ix.close()
