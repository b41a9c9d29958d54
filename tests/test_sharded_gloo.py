"""The multi-GPU host path on CPU: world_size 2 over gloo.

ShardedSearcher only moves and merges results; its local search and merge are injected.  Here both
are the CPU oracle (tests may use it), which exercises exactly the code the GPU ranks run around the
HIP calls: shard bounds, local->global ids, the packed slot layout of the ONE all-gather and the merge
contract.  The second test runs bench.py's own control flow (`bench.run`: row shards by default for
N > 1, 10M-row default, the sharded searcher, max-over-ranks timing, the rank-0 parity sample, the JSON
line) with a CPU backend in place of the HIP one.
"""
import json
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


def _oracle_local_search(orc, base, metric):
    def local_search(tq, kk):
        ids, dists, counts = orc.flat_batch_search(tq.numpy(), base, kk, metric, threads=2)
        return (torch.from_numpy(ids.view(np.int64).copy()), torch.from_numpy(dists.copy()),
                torch.from_numpy(counts.view(np.int32).copy()))

    return local_search


def _oracle_merge(orc):
    def merge(slots, world, nq, kk):
        from needle_amd.sharded import unpack_slots

        all_i, all_d, all_c = unpack_slots(slots, world, nq, kk)
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

    return merge


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
    s = ShardedSearcher(_oracle_local_search(orc, base[lo:hi], metric), _oracle_merge(orc), lo)
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

    for n in (0, 1, 7, 1000, 1_000_000, 10_000_000):
        for w in (1, 2, 3, 8):
            for align in (1, 5000):
                spans = [shard_bounds(n, w, r, align) for r in range(w)]
                assert spans[0][0] == 0 and spans[-1][1] == n
                assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
                assert all(lo <= hi for lo, hi in spans)
                assert all(lo % align == 0 for lo, _ in spans if lo < n)


def test_slot_layout_round_trip():
    from needle_amd.sharded import pack_slot, slot_bytes, unpack_slots

    q, k, world = 5, 3, 2
    assert slot_bytes(q, k) == (q * k * 12 + q * 4 + 15) // 16 * 16
    slots = torch.zeros((world, slot_bytes(q, k)), dtype=torch.uint8)
    parts = []
    for w in range(world):
        ids = torch.arange(q * k, dtype=torch.int64).view(q, k) + 100 * w
        dists = torch.rand((q, k))
        counts = torch.arange(q, dtype=torch.int32) % (k + 1)
        pack_slot(slots[w], ids, dists, counts)
        parts.append((ids, dists, counts))
    ui, ud, uc = unpack_slots(slots, world, q, k)
    for w in range(world):
        assert torch.equal(ui[w], parts[w][0]) and torch.equal(ud[w], parts[w][1]) and torch.equal(uc[w], parts[w][2])


# ---- bench.py's control flow under gloo ---------------------------------------------------------------------------
class _CpuBackend:
    """bench.HipBackend's interface with numpy rows and the oracle as local search and merge."""

    name = "cpu-test"

    def __init__(self, rank, local_rank, world):
        from oracle import oracle as orc

        self.rank, self.world, self.orc = rank, world, orc
        self.profiled = 0

    def rows(self, n, dim, seed, first_row=0):
        import bench

        assert first_row % bench.GEN_CHUNK == 0
        out = np.empty((n, dim), dtype=np.float32)
        for s in range(0, n, bench.GEN_CHUNK):
            e = min(n, s + bench.GEN_CHUNK)
            rng = np.random.default_rng(bench.generator_seed(seed, first_row + s))
            out[s:e] = rng.standard_normal((bench.GEN_CHUNK, dim)).astype(np.float32)[: e - s]
        return torch.from_numpy(np.stack([self.orc.normalized(r) for r in out]) if n else out)

    def host(self, t):
        return t.numpy()

    def flat_index(self, base, metric):
        class _Ix:
            def __init__(self, b, m):
                self.base, self.metric = b.numpy(), int(m)

            def close(self):
                pass

        return _Ix(base, metric)

    def searcher(self, index, precision, row_offset, sharded):
        from needle_amd.sharded import ShardedSearcher

        local = _oracle_local_search(self.orc, index.base, index.metric)
        if not sharded:
            return local
        return ShardedSearcher(local, _oracle_merge(self.orc), row_offset).search

    def exchange_label(self):
        return "one gloo all-gather (CPU test)"

    def profile_begin(self):
        self.profiled += 1

    def profile_end(self):
        return {"scan_ms_total": 1.0, "scan_launches": 1, "searches": 1, "refined_queries": 0, "fallback_queries": 0}

    def sync(self):
        pass

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max_over_ranks(self, x):
        t = torch.tensor([x], dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def paths(self, want, oracle):
        return []

    def close(self):
        pass


def _bench_worker(rank, world, port, argv, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    args = bench.parse(argv)
    line = bench.run(args, _CpuBackend(rank, rank, world))
    with open(os.path.join(out_dir, f"bench{rank}.json"), "w") as f:
        json.dump({"line": line, "default_rows": bench.parse(["--gpus", "2"]).rows}, f)
    dist.destroy_process_group()


def test_bench_control_flow_row_shards_by_default(tmp_path, oracle):
    """`bench.py --gpus 2` with no other flag: row blocks of a 10M-row base, one all-gather, merge; here at a size the CPU
    affords (--rows given) but through the same `run`."""
    world = 2
    argv = ["--gpus", "2", "--rows", "15000", "--dim", "24", "--queries", "12", "--k", "5", "--steps", "1", "--warmup", "0",
            "--secondary-steps", "0", "--parity-queries", "12"]
    mp.spawn(_bench_worker, args=(world, _free_port(), argv, str(tmp_path)), nprocs=world, join=True)
    rec0 = json.load(open(tmp_path / "bench0.json"))
    rec1 = json.load(open(tmp_path / "bench1.json"))
    assert rec1["line"] is None and rec0["default_rows"] is None  # rows default resolved inside run()
    line = rec0["line"]
    assert line["config"]["sharding"] == "row-blocks x2, all-gather + merge" and line["scaling"] == "strong"
    assert line["config"]["rows_total"] == 15000 and line["config"]["rows_per_gpu"] == 10000  # blocks on generator chunks
    assert line["n_gpus"] == 2 and line["config"]["queries_total"] == 12
    assert line["parity"] == {"queries_checked": 12, "rank": 0, "ids_equal": True, "dists_bit_equal": True}
    assert line["config"]["recall_at_10"] == 1.0 and line["unit"] == "queries/s" and line["value"] > 0


def test_bench_two_workloads_at_n_gt_1(tmp_path, oracle):
    """`bench.py --gpus 2` without --rows: `value` comes from the N = 1 workload split by rows, the larger base is measured
    by the same code and arrives as the `flat10m_sharded` entry of `paths`, with its own parity sample."""
    world = 2
    argv = ["--gpus", "2", "--headline-rows", "10000", "--big-rows", "25000", "--dim", "24", "--queries", "12", "--k", "5",
            "--steps", "1", "--warmup", "0", "--secondary-steps", "0", "--parity-queries", "12"]
    mp.spawn(_bench_worker, args=(world, _free_port(), argv, str(tmp_path)), nprocs=world, join=True)
    line = json.load(open(tmp_path / "bench0.json"))["line"]
    assert json.load(open(tmp_path / "bench1.json"))["line"] is None
    assert line["config"]["rows_total"] == 10000 and line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["parity"]["ids_equal"] and line["parity"]["dists_bit_equal"]
    assert line["config"]["single_gpu_same_workload"]["value"] > 0
    (big,) = line["paths"]
    assert big["path"] == "flat10m_sharded" and big["rows_total"] == 25000 and big["rows_per_gpu"] == 15000
    assert big["parity"] == {"queries_checked": 12, "rank": 0, "ids_equal": True, "dists_bit_equal": True}
    assert big["single_gpu_same_workload"]["value"] > 0 and big["qps"] > 0
    import bench

    assert bench.parity_ok(line)
    big["parity"]["ids_equal"] = False
    assert not bench.parity_ok(line)


def test_bench_defaults():
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse([])
    assert a.gpus == 1 and a.rows is None and a.shard is None and a.precision == "bf16"
    import inspect

    # `value` is measured on the metric's 1M-row base at every N; at N > 1 its 10M-row base follows as `paths` entries
    assert a.headline_rows == 1_000_000 and a.big_rows == 10_000_000
    src = inspect.getsource(bench.run)
    assert 'run_workload(args, be, args.headline_rows, "headline")' in src and 'run_workload(args, be, args.big_rows, "big"' in src
    assert not bench.parity_ok({"parity": {"ids_equal": False, "dists_bit_equal": True}})
    assert bench.parity_ok({"parity": None, "paths": [{"parity": {"ids_equal": True, "queries_checked": 0}}]})


def test_bench_starts_its_own_ranks(capfd, monkeypatch, oracle):
    """`python bench.py --gpus 2` with no launcher around it: the script starts two ranks itself (child processes with the
    launcher's environment), rank 0's line says n_gpus 2, and a failing rank fails the run."""
    sys.path.insert(0, ROOT)
    import bench

    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(key, raising=False)
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    argv = ["--gpus", "2", "--rows", "15000", "--dim", "24", "--queries", "12", "--k", "5", "--steps", "1", "--warmup", "0",
            "--secondary-steps", "0", "--parity-queries", "12", "--backend", "test_sharded_gloo:_CpuBackend"]
    capfd.readouterr()
    assert bench.main(argv) == 0
    out = capfd.readouterr().out
    lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out  # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "row-blocks x2, all-gather + merge"
    assert line["parity"]["ids_equal"] and line["parity"]["dists_bit_equal"]
    # a rank that cannot start ends the whole run with a non-zero status
    assert bench.main(argv[:-1] + ["test_sharded_gloo:_NoSuchBackend"]) != 0
    # the product wiring without enough GPUs refuses instead of reporting one rank as two
    if torch.cuda.device_count() < 2:
        assert bench.main(["--gpus", "2"]) == 2
    # one rank told to call itself two GPUs refuses as well
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.main(["--gpus", "2", "--backend", "test_sharded_gloo:_CpuBackend"]) == 2
