"""The multi-GPU host path on CPU: world_size 2 over gloo.

ShardedSearcher only moves and merges results; its local search and merge are injected.  Here both
are the CPU oracle (tests may use it), which exercises exactly the code the GPU ranks run around the
HIP calls: shard bounds, local->global ids, the all-gather layout [world, q, k] and the merge contract.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, q, k, metric, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from needle_amd.sharded import ShardedSearcher, shard_bounds
    from oracle import oracle as orc

    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[n // 2 + 3] = base[1]  # a duplicate across the shard boundary: tie broken by global id
    queries = rng.standard_normal((q, d)).astype(np.float32)
    queries[0] = base[1]
    lo, hi = shard_bounds(n, world, rank)

    def local_search(tq, kk):
        ids, dists, counts = orc.flat_batch_search(tq.numpy(), base[lo:hi], kk, metric, threads=2)
        return (torch.from_numpy(ids.view(np.int64).copy()), torch.from_numpy(dists.copy()),
                torch.from_numpy(counts.view(np.int32).copy()))

    def merge(all_d, all_i, all_c, kk):
        w, nq, per = all_i.shape
        oi = np.full((nq, kk), -1, dtype=np.int64)
        od = np.full((nq, kk), np.inf, dtype=np.float32)
        oc = np.zeros(nq, dtype=np.int32)
        for j in range(nq):
            mi, md = orc.merge_topk(all_d[:, j, :].numpy(), all_i[:, j, :].numpy().view(np.uint64),
                                    all_c[:, j].numpy().view(np.uint32), kk)
            oi[j, :len(mi)] = mi.view(np.int64)
            od[j, :len(md)] = md
            oc[j] = len(mi)
        return torch.from_numpy(oi), torch.from_numpy(od), torch.from_numpy(oc)

    s = ShardedSearcher(local_search, merge, lo)
    ids, dists, counts = s.search(torch.from_numpy(queries), k)
    gi, gd, gc = orc.flat_batch_search(queries, base, k, metric, threads=2)
    ok = (np.array_equal(ids.numpy().view(np.uint64), gi) and np.array_equal(dists.numpy().view(np.uint32),
                                                                            gd.view(np.uint32))
          and np.array_equal(counts.numpy().view(np.uint32), gc))
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else f"mismatch\n{ids[:2]}\n{gi[:2]}")
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [2, 0])  # Euclidean, Cosine
def test_row_sharded_search_matches_unsharded(tmp_path, metric, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 1001, 24, 9, 7, metric, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_shard_bounds_cover_everything():
    from needle_amd.sharded import shard_bounds

    for n in (0, 1, 7, 1000, 1_000_000):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert all(lo <= hi for lo, hi in spans)
