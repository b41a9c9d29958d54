"""The flat search at the benchmark's full sizes (1M x 384 cosine, 10K queries, k = 10; 1M x 768 dot product under a
10 % filter, k = 100), where the CPU oracle can only afford a sample.  Beside that sample the results are held to
properties that do not depend on the size: a repeated search and a search with the other first-pass precision return
the same bits (both are proofs of the same exact answer); every query's hits are k distinct live rows in ascending
(distance, id) order; and the index cut into two shards, searched separately and merged -- the multi-GPU sequence --
returns the whole index's result bit for bit."""
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


def _search(ix, dq, k, precision, d_filter=None):
    import torch

    nq = dq.shape[0]
    ids = torch.empty((nq, k), dtype=torch.int64, device=dq.device)
    dists = torch.empty((nq, k), dtype=torch.float32, device=dq.device)
    counts = torch.empty((nq,), dtype=torch.int32, device=dq.device)
    torch.cuda.synchronize()
    ix.search_device(dq, k, ids, dists, counts, precision=precision, d_filter=d_filter)
    torch.cuda.synchronize()
    return ids, dists, counts


def _same(a, b):
    import torch

    return all(torch.equal(x, y) for x, y in zip(a, b))  # float tensors hold no NaN here: equal values = equal bits


def _ordered(ids, dists):
    d0, d1 = dists[:, :-1], dists[:, 1:]
    return bool(((d0 < d1) | ((d0 == d1) & (ids[:, :-1] < ids[:, 1:]))).all())


def test_headline_shape_properties_and_sample(ctx, oracle):
    import torch

    import needle_amd as na

    dev = torch.device("cuda", 0)
    n, dim, nq, k = 1_000_000, 384, 10_000, 10
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    base = torch.nn.functional.normalize(torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32), dim=1)
    queries = torch.nn.functional.normalize(torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32), dim=1)
    queries[:64] = base[::15625][:64]  # planted: these rows must find themselves
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    first = _search(ix, queries, k, "bf16")
    ids, dists, counts = first
    assert bool((counts == k).all())
    assert _ordered(ids, dists)
    assert bool((ids >= 0).all()) and bool((ids < n).all())
    assert torch.equal(ids[:64, 0].cpu(), torch.arange(0, n, 15625)[:64])
    assert bool((dists[:64, 0].abs() < 1e-6).all())
    assert _same(first, _search(ix, queries, k, "bf16"))          # idempotent, adaptive state included
    fp32 = _search(ix, queries[:2000], k, "fp32")                 # the other first pass proves the same answer
    assert _same((ids[:2000], dists[:2000], counts[:2000]), fp32)
    ix.close()

    # two shards searched separately, ids offset, merged: the whole index's result
    half = n // 2
    parts = []
    for s in range(2):
        sh = na.FlatIndex(ctx, base[s * half:(s + 1) * half], na.DistanceFunction.Cosine)
        pi, pd, pc = _search(sh, queries, k, "bf16")
        parts.append((pi + s * half, pd, pc))
        sh.close()
    md = torch.stack([p[1] for p in parts]).contiguous()
    mi = torch.stack([p[0] for p in parts]).contiguous()
    mc = torch.stack([p[2] for p in parts]).contiguous()
    oi, od, oc = torch.empty_like(ids), torch.empty_like(dists), torch.empty_like(counts)
    torch.cuda.synchronize()
    ctx.merge_topk_device(md, mi, mc, 2, nq, k, k, oi, od, oc)
    torch.cuda.synchronize()
    assert _same(first, (oi, od, oc))

    # the oracle's sample: planted and random queries
    pick = [0, 1, 63, 64, 65, 5000, 9998, 9999]
    hb = base.cpu().numpy()
    hq = queries[pick].cpu().numpy()
    wi, wd, wc = oracle.flat_batch_search(hq, hb, k, oracle.COSINE)
    gi = ids[pick].cpu().numpy().view(np.uint64)
    gd = dists[pick].cpu().numpy()
    assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
    assert np.array_equal(counts[pick].cpu().numpy().view(np.uint32), wc)


def test_filtered_dot_product_shape_properties_and_sample(ctx, oracle):
    import torch

    import needle_amd as na
    from needle_amd.index import _mask_words

    dev = torch.device("cuda", 0)
    n, dim, nq, k = 1_000_000, 768, 1_000, 100
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    base = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
    queries = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
    keep = np.random.default_rng(7).random(n) < 0.1
    d_filter = torch.from_numpy(_mask_words(keep, n).view(np.int32)).to(dev)
    d_keep = torch.from_numpy(keep).to(dev)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.DotProduct)
    first = _search(ix, queries, k, "bf16", d_filter)
    ids, dists, counts = first
    assert bool((counts == k).all())
    assert _ordered(ids, dists)
    assert bool(d_keep[ids.reshape(-1)].all())                    # only rows the filter admits
    assert _same(first, _search(ix, queries, k, "bf16", d_filter))
    assert _same(first, _search(ix, queries, k, "fp32", d_filter))
    ix.close()
    pick = [0, 499, 999]
    wi, wd, wc = oracle.flat_batch_search(queries[pick].cpu().numpy(), base.cpu().numpy(), k, oracle.DOT_PRODUCT, filt=keep)
    assert np.array_equal(ids[pick].cpu().numpy().view(np.uint64), wi)
    assert np.array_equal(dists[pick].cpu().numpy().view(np.uint32), wd.view(np.uint32))
