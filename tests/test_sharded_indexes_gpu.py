"""Row-sharded IVF and SQ8 indexes: every shard holds its row block's slice of the inverted lists / code matrix with the
centroids / quantizer replicated; the shards' top-k lists, ids shifted to global, merged by (distance, id) through
needle_hip_merge_topk_device reproduce the whole index's result (the oracle's), bit for bit.  One process plays both
ranks: the exchange itself (all-gather) is covered on CPU by test_sharded_gloo.py."""
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


def _merged(ctx, searches, offsets, queries, k):
    import torch

    from needle_amd.sharded import hip_merge

    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(queries).to(dev)
    parts = []
    for run, off in zip(searches, offsets):
        ids, dists, counts = run(dq, k)
        parts.append((torch.where(ids >= 0, ids + off, ids), dists, counts))  # ShardedSearcher.search's shift
    ai = torch.stack([p[0] for p in parts]).contiguous()
    ad = torch.stack([p[1] for p in parts]).contiguous()
    ac = torch.stack([p[2] for p in parts]).contiguous()
    ids, dists, counts = hip_merge(ctx)(ad, ai, ac, k)
    return ids.cpu().numpy().view(np.uint64), dists.cpu().numpy(), counts.cpu().numpy().view(np.uint32)


def _check(got, want):
    ids, dists, counts = got
    oi, od, oc = want
    assert np.array_equal(counts, oc)
    for i in range(ids.shape[0]):
        c = int(oc[i])
        assert np.array_equal(ids[i, :c], oi[i, :c]), (i, ids[i, :c], oi[i, :c])
        assert np.array_equal(dists[i, :c].view(np.uint32), od[i, :c].view(np.uint32)), i


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_ivf_shards_equal_the_whole_index(ctx, oracle, precision):
    from needle_amd.ivf import IvfConfig, IvfIndex
    from needle_amd.sharded import hip_local_search_handle, shard_bounds

    rng = np.random.default_rng(21)
    n, d, nlist, nprobe, k = 30001, 64, 48, 7, 10  # odd n: unequal blocks
    base = rng.standard_normal((n, d)).astype(F)
    queries = rng.standard_normal((120, d)).astype(F)
    cents = oracle.ivf_kmeans(base[:2000], nlist, max_iter=5, seed=3)
    assign = oracle.ivf_assign(base, cents)
    offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
    want = oracle.ivf_batch_search(queries, base, cents, offsets, order, nprobe, k)
    shards, runs, offs = [], [], []
    for rank in range(2):
        lo, hi = shard_bounds(n, 2, rank)
        ix = IvfIndex(d, IvfConfig.new(nlist).with_nprobe(nprobe), ctx=ctx)
        ix.set_centroids(cents)  # replicated: identical probe lists on every rank
        ix.add(range(hi - lo), base[lo:hi])
        shards.append(ix)
        runs.append(hip_local_search_handle(ix._resident(), nprobe=nprobe, precision=precision))
        offs.append(lo)
    _check(_merged(ctx, runs, offs, queries, k), want)
    for ix in shards:
        ix.close()


def test_two_sq8_shards_equal_the_whole_index(ctx, oracle):
    from needle_amd.quantization import ScalarQuantizer, Sq8Index
    from needle_amd.sharded import hip_local_search_handle, shard_bounds

    rng = np.random.default_rng(22)
    n, d, k = 20001, 96, 10
    base = rng.standard_normal((n, d)).astype(F)
    queries = rng.standard_normal((50, d)).astype(F)
    osq = oracle.ScalarQuantizer.train(base[:2000])
    codes = osq.quantize(base)
    want = osq.batch_search(queries, codes, k)
    sq = ScalarQuantizer(osq.min_vals, osq.max_vals, osq.scale)  # replicated quantizer parameters
    shards, runs, offs = [], [], []
    for rank in range(2):
        lo, hi = shard_bounds(n, 2, rank)
        ix = Sq8Index(ctx, sq, codes[lo:hi])
        shards.append(ix)
        runs.append(hip_local_search_handle(ix._h, precision="bf16"))
        offs.append(lo)
    _check(_merged(ctx, runs, offs, queries, k), want)
    for ix in shards:
        ix.close()
