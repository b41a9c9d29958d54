"""Re-entrancy of the batch search: several host threads search ONE index at the same time (the reference calls
Collection::batch_search under a read lock from many threads, src/database/internal.rs:319).  Every thread must get
exactly what a lone call gets, for flat, filtered, IVF and SQ8 indexes, and the context's metrics must add up."""
import threading

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


def _hammer(search, batches, rounds=3):
    """search(batch_index) from len(batches) threads at once, `rounds` times each; returns results[thread][round]."""
    out = [[None] * rounds for _ in batches]
    errs = []
    start = threading.Barrier(len(batches))

    def work(t):
        try:
            start.wait()
            for r in range(rounds):
                out[t][r] = search(t)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(batches))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    return out


def _equal(a, b):
    return all(np.array_equal(x.view(np.uint32) if x.dtype == F else x, y.view(np.uint32) if y.dtype == F else y)
               for x, y in zip(a, b))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_concurrent_flat_searches(ctx, oracle, precision):
    import needle_amd as na

    rng = np.random.default_rng(8)
    base = rng.standard_normal((60_000, 96)).astype(F)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    batches = [rng.standard_normal((300 + 50 * t, 96)).astype(F) for t in range(6)]  # more threads than slots
    keep = rng.random(60_000) < 0.3
    serial = [ix.search(b, 10, precision=precision) for b in batches]
    serial_f = [ix.search(b, 10, precision=precision, filter_mask=keep) for b in batches]
    ctx.reset_metrics()
    got = _hammer(lambda t: ix.search(batches[t], 10, precision=precision), batches)
    m = ctx.metrics()
    assert m["searches"] == 6 * 3 and m["queries"] == 3 * sum(b.shape[0] for b in batches)
    for t in range(6):
        for r in range(3):
            assert _equal(got[t][r], serial[t]), (t, r)
    gotf = _hammer(lambda t: ix.search(batches[t], 10, precision=precision, filter_mask=keep), batches)
    for t in range(6):
        assert _equal(gotf[t][0], serial_f[t]) and _equal(gotf[t][2], serial_f[t]), t
    oi, od, oc = oracle.flat_batch_search(batches[0][:20], base, 10, oracle.COSINE)
    assert np.array_equal(serial[0][0][:20], oi) and np.array_equal(serial[0][1][:20].view(np.uint32), od.view(np.uint32))
    ix.close()


def test_concurrent_ivf_and_sq8_searches(ctx, oracle):
    from needle_amd.ivf import IvfConfig, IvfIndex
    from needle_amd.quantization import ScalarQuantizer, Sq8Index

    rng = np.random.default_rng(9)
    n, d, nlist, nprobe = 40_000, 64, 64, 8
    base = rng.standard_normal((n, d)).astype(F)
    cents = oracle.ivf_kmeans(base[:2000], nlist, max_iter=5, seed=3)
    ivf = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe), ctx=ctx)
    ivf.set_centroids(cents)
    ivf.add(range(n), base)
    batches = [rng.standard_normal((200 + 30 * t, d)).astype(F) for t in range(5)]
    serial = [ivf.batch_search_arrays(b, 10, precision="bf16") for b in batches]
    got = _hammer(lambda t: ivf.batch_search_arrays(batches[t], 10, precision="bf16"), batches)
    for t in range(5):
        for r in range(3):
            assert _equal(got[t][r], serial[t]), (t, r)
    ivf.close()
    osq = oracle.ScalarQuantizer.train(base[:2000])
    codes = osq.quantize(base)
    sq = Sq8Index(ctx, ScalarQuantizer(osq.min_vals, osq.max_vals, osq.scale), codes)
    serial = [sq.search(b, 10, precision="bf16") for b in batches]
    got = _hammer(lambda t: sq.search(batches[t], 10, precision="bf16"), batches)
    for t in range(5):
        assert _equal(got[t][0], serial[t]) and _equal(got[t][2], serial[t]), t
    sq.close()


def test_stale_handles_never_name_a_newer_object(oracle):
    """A wrapper that outlives its context (garbage collected late, say) hands a dead handle to the destroy call: that
    must stay a no-op however many indexes were created since -- addresses of destroyed handles are not reused."""
    import ctypes as C

    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(3)
    base = rng.standard_normal((500, 16)).astype(F)
    q = rng.standard_normal((4, 16)).astype(F)
    stale = []
    for _ in range(8):
        c = na.Context(0)
        ix = na.FlatIndex(c, base, na.DistanceFunction.Euclidean)
        stale.append((C.c_void_p(ix._h.value), C.c_void_p(c._h.value)))
        c.close()       # takes the index along
        ix.close()      # wrapper of a closed context: nothing reaches the library
    c = na.Context(0)
    live = [na.FlatIndex(c, base, na.DistanceFunction.Euclidean) for _ in range(8)]
    assert not {ix._h.value for ix in live} & {h.value for h, _ in stale}
    for h, hc in stale:  # what a late __del__ of an unguarded binding would do
        _lib.lib().needle_hip_index_destroy(h)
        _lib.lib().needle_hip_shutdown(hc)
    want = oracle.flat_batch_search(q, base, 5, oracle.EUCLIDEAN)
    for ix in live:
        got = ix.search(q, 5)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        ix.close()
    c.close()
