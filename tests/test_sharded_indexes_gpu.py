"""Row-sharded IVF and SQ8 indexes: every shard holds its row block's slice of the inverted lists / code matrix with the
centroids / quantizer replicated; the shards' top-k lists, ids shifted to global, merged by (distance, id) through
needle_hip_merge_shard_slots_device (the merge kernel of the RCCL path) reproduce the whole index's result (the oracle's), bit for bit.  One process plays both
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

    from needle_amd.sharded import hip_merge, pack_slot, slot_bytes

    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(queries).to(dev)
    q = queries.shape[0]
    slots = torch.zeros((len(searches), slot_bytes(q, k)), dtype=torch.uint8, device=dev)  # what the all-gather delivers
    for w, (run, off) in enumerate(zip(searches, offsets)):
        ids, dists, counts = run(dq, k)
        pack_slot(slots[w], torch.where(ids >= 0, ids + off, ids), dists, counts)  # ShardedSearcher.search's shift
    ids, dists, counts = hip_merge(ctx)(slots, len(searches), q, k)
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


@pytest.mark.parametrize("k", [10, 100])
def test_eight_flat_shards_equal_the_whole_index(ctx, oracle, k):
    """The 8-way merge of the full node: eight row blocks of one base searched separately (k = 10: the one-wave-per-query
    form of the merge kernel, 80 entries; k = 100: the workgroup-per-query form, 800 entries), merged == the whole."""
    import needle_amd as na
    from needle_amd.sharded import hip_local_search, shard_bounds

    rng = np.random.default_rng(23)
    n, d = 40_003, 48
    base = rng.standard_normal((n, d)).astype(F)
    base[n - 1] = base[17]          # the same row in the first and in the last block: tie broken by global id
    base[3 * (n // 8) + 5] = base[17]
    queries = rng.standard_normal((203, d)).astype(F)  # not a multiple of the four queries a workgroup merges
    queries[0] = base[17]
    want = oracle.flat_batch_search(queries, base, k, oracle.COSINE)
    shards, runs, offs = [], [], []
    for rank in range(8):
        lo, hi = shard_bounds(n, 8, rank)
        ix = na.FlatIndex(ctx, base[lo:hi], na.DistanceFunction.Cosine)
        shards.append(ix)
        runs.append(hip_local_search(ix, "bf16"))
        offs.append(lo)
    _check(_merged(ctx, runs, offs, queries, k), want)
    for ix in shards:
        ix.close()


def test_rccl_entry_point_world_of_one(ctx, oracle):
    """needle_hip_sharded_batch_search_device with a one-rank RCCL communicator: local search, id offset, the (trivial)
    all-gather and the merge kernel all run; the result is the shard's own, ids shifted by row_offset."""
    import torch

    import needle_amd as na
    from needle_amd.sharded import Communicator, RcclShardedSearcher

    rng = np.random.default_rng(31)
    base = rng.standard_normal((5000, 72)).astype(F)
    base[77] = base[5]  # a tie inside the shard
    queries = rng.standard_normal((130, 72)).astype(F)
    queries[0] = base[5]
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    comm = Communicator(ctx)
    assert comm.world == 1 and comm.rank == 0
    dq = torch.from_numpy(queries).to(torch.device("cuda", 0))
    torch.cuda.synchronize()
    for k, off in ((10, 0), (7, 123_000_000_000), (0, 5)):
        ids, dists, counts = RcclShardedSearcher(ix._h, comm, off, precision="bf16").search(dq, k)
        wi, wd, wc = oracle.flat_batch_search(queries, base, k, oracle.EUCLIDEAN)
        assert np.array_equal(counts.cpu().numpy().view(np.uint32), wc)
        if k:
            assert np.array_equal(ids.cpu().numpy().view(np.uint64), wi + np.uint64(off))
            assert np.array_equal(dists.cpu().numpy().view(np.uint32), wd.view(np.uint32))
    comm.close()
    ix.close()
