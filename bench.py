#!/usr/bin/env python3
"""Batch-search throughput of the exact flat scan on MI355X.

A "step" is one complete batch search: Q queries against the resident base, ids and distances of the
k nearest per query delivered (first pass on the matrix cores, candidate merge, reference-order
re-scoring, bound check, exact scan for rejected queries; with more than one GPU also the all-gather
of per-shard results and the merge).  Inputs are resident in HBM before the timed region starts; the
index build (upload, norms, operand copy) is outside it, like the reference's insert phase.

--gpus 1 (default): exact flat batch_search, 1M x 384 cosine, 10K queries, k=10, unit-normalised Gaussian
embeddings, bf16 first pass (the delivered ids and distances are those of the reference's f32 scalar
arithmetic whatever the first pass; the fp32 first pass is timed beside it).  The JSON line also carries
`paths`: the other configurations at their per-GPU share (10M x 384 flat, filtered 1M x 768 dot product,
IVF nlist 4096 / nprobe 32 over 1.25M x 384, SQ8 and PQ m=48 over 6.25M x 384), each with its own roofline
object, parity sample and host-pointer rate, and the reference's HNSW restated on the CPU over a bounded sample.

--gpus N > 1 (one rank per GPU: under torch.distributed.run, or started by this script itself when no launcher set
WORLD_SIZE -- never a silent single-GPU run): the base is split into N contiguous row blocks, every GPU scans its
block for ALL queries, the per-shard results travel in one RCCL all-gather and are merged on every rank
(needle_hip_sharded_batch_search_device): strong scaling of one batch.  `value` is measured on the SAME workload as at
N = 1 (1M x 384, 10K queries), so the values of N = 1, 2, 4, 8 form one strong-scaling curve; the metric's 10M x 384
base, split the same way, follows in the same run as `paths` entries (`flat10m_sharded`: exact flat; `ivf_sharded`:
nlist 4096 / nprobe 32), each with the one-GPU time of its own workload measured by rank 0 beside it.  `--rows R`
runs only an R-row base; `--shard queries` is the opt-in alternative (every GPU holds the whole base and serves its
own queries, no exchange).

NEEDLE_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks: ranks
share devices, the one all-gather goes over gloo and the merge runs through needle_hip_merge_shard_slots_device.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense, MI355X
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense
BOOST_CLOCK_GHZ = 2.4           # the clock both peaks are quoted at
KERNEL_SOURCES = {"flat_scan_bf16": ("flat_scan_bf16.hip", "nh_scan.h", "nh_select.h"),
                  "flat_scan_f32": ("flat_scan.hip", "nh_scan.h", "nh_select.h")}


def kernel_source_hash(kernel):
    """sha256 over the sources of a first-pass kernel (what tools/summarize_prof.py stamps a counter record with)."""
    import hashlib

    h = hashlib.sha256()
    try:
        for name in KERNEL_SOURCES[kernel]:
            with open(os.path.join(ROOT, "needle_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
    except (OSError, KeyError):
        return None
    return h.hexdigest()
GEN_CHUNK = 5000                # rows per generator stream; shard starts are multiples of it
METRIC_LABEL = "batch-search QPS @ recall@10, 1M & 10M×384 cosine k=10, 1/2/4/8 MI355X"
DEFAULT_PATHS = "flat10m,c5,ivf,ivf_clustered,sq8,pq,hnsw"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=None,
                    help="total base rows (default: 1M, split across the GPUs; at N > 1 the 10M-row base follows as `paths` entries)")
    ap.add_argument("--big-rows", type=int, default=10_000_000,
                    help="N > 1, --rows not given: rows of the second, larger base (0 = skip it)")
    ap.add_argument("--headline-rows", type=int, default=1_000_000,
                    help="rows of the base `value` is measured on when --rows is not given (the metric's 1M; smaller in tests)")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--shard", default=None, choices=["queries", "rows"],
                    help="N > 1: rows (default; row blocks, all-gather + merge) or queries (replicas, no exchange)")
    ap.add_argument("--secondary-steps", type=int, default=2, help="steps of the other first-pass precision (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--ivf-leg-seconds", type=float, default=420.0, help="N > 1: time limit of the row-sharded IVF leg")
    ap.add_argument("--parity-queries", type=int, default=16, help="N > 1: queries rank 0 checks against the oracle (0 = skip)")
    ap.add_argument("--paths", default=DEFAULT_PATHS, help="N = 1: other configurations folded into the line ('none' = skip)")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="N > 1: do not let rank 0 also search the whole base on its own GPU (config.single_gpu_same_workload)")
    ap.add_argument("--hnsw-rows", type=int, default=100_000, help="rows of the CPU HNSW baseline in `paths` (sequential build)")
    ap.add_argument("--hnsw-seconds", type=float, default=150.0,
                    help="time budget of that build: it stops there and the baseline covers the rows that are in")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per first-pass launch from a separate rocprofv3 --pmc run")
    ap.add_argument("--backend", default="hip",
                    help="'hip' (the product wiring) or module:Class of a stand-in with HipBackend's interface (CPU tests)")
    return ap.parse_args(argv)


def generator_seed(seed, first_row):
    return seed * 1_000_003 + first_row // GEN_CHUNK


class HipBackend:
    """The product wiring: device tensors, libneedle_hip through the C ABI, RCCL (or a gloo rehearsal) between ranks."""

    name = "hip"

    def __init__(self, rank, local_rank, world):
        self.rank, self.world = rank, world
        # gloo: ranks share devices, timing reduced over CPU tensors; the measured runs use RCCL, one rank per GPU
        self.transport = os.environ.get("NEEDLE_BENCH_BACKEND", "nccl")
        dev_index = local_rank if self.transport == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        if world > 1 and not dist.is_initialized():
            if self.transport == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(self.transport)
        import needle_amd as na

        self.na = na
        self.ctx = na.Context(dev_index)
        # A/B sweeps of the harness (tools/ab.sh): planning overrides handed to needle_hip_set_tuning, e.g. "stages=15:30"
        self.ctx.apply_tuning_spec(os.environ.get("NEEDLE_BENCH_TUNE"))
        self.comm = None
        self.comm_error = None

    # ---- data -----------------------------------------------------------------------------------
    def rows(self, n, dim, seed, first_row=0):
        """Unit-normalised Gaussian rows; row i depends only on (seed, i), so shards agree with the whole.  Normalised with
        the reference's recipe (src/distance.rs:426-434) by needle_hip_normalize_device."""
        assert first_row % GEN_CHUNK == 0, "shard boundaries must fall on generator chunks"
        out = torch.empty((n, dim), dtype=torch.float32, device=self.device)
        for s in range(0, n, GEN_CHUNK):
            e = min(n, s + GEN_CHUNK)
            g = torch.Generator(device=self.device)
            g.manual_seed(generator_seed(seed, first_row + s))
            out[s:e] = torch.randn((GEN_CHUNK, dim), generator=g, device=self.device, dtype=torch.float32)[: e - s]
        torch.cuda.synchronize()
        self.ctx.normalize_device(out)
        return out

    def host(self, t):
        return t.cpu().numpy()

    # ---- index and search -------------------------------------------------------------------------
    def flat_index(self, base, metric):
        return self.na.FlatIndex(self.ctx, base, metric)

    def searcher(self, index, precision, row_offset, sharded):
        from needle_amd.sharded import (Communicator, RcclShardedSearcher, ShardedSearcher, hip_local_search, hip_merge)

        if not sharded:
            return hip_local_search(index, precision)
        if self.transport == "nccl" and self.comm is None and self.comm_error is None:
            # the communicator behind the C ABI; every rank must end up on the same exchange, so the outcome is agreed on
            try:
                self.comm = Communicator(self.ctx)
                ok = 1
            except Exception as e:  # noqa: BLE001
                self.comm_error, ok = f"{type(e).__name__}: {e}", 0
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if self.comm is not None:
                    self.comm.close()
                    self.comm = None
                self.comm_error = self.comm_error or "needle_hip_comm_init failed on another rank"
                print(f"[bench] rank {self.rank}: {self.comm_error}; exchanging through torch.distributed", file=sys.stderr)
        if self.transport == "nccl" and self.comm is not None:
            return RcclShardedSearcher(index._h, self.comm, row_offset, precision=precision).search
        return ShardedSearcher(hip_local_search(index, precision), hip_merge(self.ctx), row_offset).search

    def exchange_label(self):
        if self.transport != "nccl":
            return f"one {self.transport} all-gather (rehearsal), device merge"
        if self.comm_error is not None:
            return "one RCCL all-gather through torch.distributed, device merge (C-ABI communicator unavailable)"
        return "one RCCL all-gather behind the C ABI"

    def profile_begin(self):
        self.ctx.set_profiling(True)
        self.ctx.reset_metrics()

    def profile_end(self):
        m = self.ctx.metrics()
        self.ctx.set_profiling(False)
        return m

    # ---- ranks ------------------------------------------------------------------------------------
    def sync(self):
        torch.cuda.synchronize()

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.transport == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def host_pointer_qps(self, index, queries_host, k, precision):
        import tools.bench_paths as bp

        return bp.host_api_qps(index._h, queries_host, k, precision=precision)

    def ivf_sharded(self, base, queries, row_offset, rows_total, k, exact_ids, steps):
        """The metric's IVF configuration (nlist 4096, nprobe 32) over the same row blocks: every rank builds an inverted
        index over ITS rows with the SAME centroids (trained by rank 0 on the head of its block, broadcast), so the probe
        lists agree on every rank and the merged result is the whole index's.  Timed like the flat leg; recall@10 is
        taken against the exact result of the flat leg."""
        import ctypes as C

        import tools.bench_paths as bp
        from needle_amd import _lib
        from needle_amd.sharded import RcclShardedSearcher, ShardedSearcher, hip_local_search_handle, hip_merge

        nlist, nprobe = 4096, 32
        n, dim = int(base.shape[0]), int(base.shape[1])
        cents = torch.zeros((nlist, dim), dtype=torch.float32)
        train_s = 0.0
        if self.rank == 0:
            sample = base[: min(n, 100 * nlist)].cpu().numpy()
            c, train_s = bp.trained_centroids(self.ctx, sample, nlist, 8)
            cents = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
        if self.world > 1:
            t = cents.to(self.device) if self.transport == "nccl" else cents
            dist.broadcast(t, src=0)
            cents = t.cpu()
        cents_h = np.ascontiguousarray(cents.numpy())
        hb = base.cpu().numpy()
        h = C.c_void_p()
        t0 = time.perf_counter()
        _lib.check(_lib.lib().needle_hip_ivf_build(self.ctx._h, hb.ctypes.data, n, dim, cents_h.ctypes.data, nlist, None, None,
                                                   C.byref(h)))
        build_s = time.perf_counter() - t0
        del hb
        try:
            if self.transport == "nccl" and self.comm is not None:
                search = RcclShardedSearcher(h, self.comm, row_offset, nprobe=nprobe, precision="bf16").search
            else:
                search = ShardedSearcher(hip_local_search_handle(h, nprobe, "bf16"), hip_merge(self.ctx), row_offset).search
            out = search(queries, k)
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = search(queries, k)
            self.barrier()
            el = self.max_over_ranks(time.perf_counter() - t0)
        finally:
            _lib.lib().needle_hip_index_destroy(h)
        got, want = out[0].cpu().numpy(), exact_ids.cpu().numpy()
        hits = sum(len(np.intersect1d(got[i], want[i])) for i in range(got.shape[0]))
        return {"path": "ivf_sharded",
                "workload": f"IVF-Flat nlist={nlist} nprobe={nprobe} over {rows_total}x{dim} unit rows in {self.world} row blocks "
                            f"(same centroids on every rank), k={k}",
                "queries": int(queries.shape[0]), "ms_per_batch": round(el / steps * 1e3, 3),
                "qps": round(queries.shape[0] * steps / el, 1), "recall_at_10_vs_exact": round(hits / max(want.size, 1), 4),
                "recall_note": "of the EXACT top 10; i.i.d. Gaussian rows have no cluster structure, so 32 of 4096 lists hold few of "
                               "them -- the returned rows are the reference IvfIndex's own (bit-equal in tests/test_config_sizes_gpu.py, "
                               "tests/test_sharded_indexes_gpu.py)",
                "exchange": self.exchange_label(), "rows_per_gpu": n, "build_s": round(build_s, 2), "train_s": round(train_s, 2)}

    def paths(self, want, oracle):
        import tools.bench_paths as bp

        return bp.run_paths(self.ctx, self.device, want, steps=2, oracle=oracle, hnsw_rows=getattr(self, "hnsw_rows", 100_000),
                            hnsw_budget_s=getattr(self, "hnsw_budget_s", 150.0))

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.ctx.close()
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def oracle_build_flags():
    """The compiler flags the CPU baseline was built with, read from its Makefile."""
    flags = "unknown"
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "Makefile")) as f:
            for ln in f:
                if ln.startswith("CFLAGS"):
                    flags = ln.split("=", 1)[1].strip()
                    break
    except OSError:
        pass
    return f"cc {flags} (no -march=native: the reference's default build targets baseline x86-64)"


def load_oracle():
    from oracle import oracle as orc  # checker + CPU baseline only; never on the product path

    orc.build()
    return orc


def oracle_sharded_truth(be, orc, args, metric, queries_host, n_shards_rows):
    """Exact top-k of `queries_host` over the whole base, the oracle scanning one row block at a time (rank 0
    regenerates every block from the generator) and its merge joining the blocks."""
    k = args.k
    per_i, per_d, per_c = [], [], []
    for lo, hi in n_shards_rows:
        blk = be.host(be.rows(hi - lo, args.dim, 42, first_row=lo))
        i, d, c = orc.flat_batch_search(queries_host, blk, k, int(metric), threads=min(orc.cpu_threads(), 16))
        per_i.append(np.where(i == np.iinfo(np.uint64).max, i, i + np.uint64(lo)))
        per_d.append(d)
        per_c.append(c)
    nq = queries_host.shape[0]
    oi = np.full((nq, k), np.iinfo(np.uint64).max, dtype=np.uint64)
    od = np.full((nq, k), np.inf, dtype=np.float32)
    oc = np.zeros(nq, dtype=np.uint32)
    for j in range(nq):
        mi, md = orc.merge_topk(np.stack([d[j] for d in per_d]), np.stack([i[j] for i in per_i]),
                                np.array([c[j] for c in per_c], dtype=np.uint32), k)
        oi[j, : mi.size], od[j, : mi.size], oc[j] = mi, md, mi.size
    return oi, od, oc


def run(args, be):
    """The benchmark's control flow on backend `be`; returns the JSON line on rank 0 (None elsewhere).

    One GPU, or --rows given: one workload.  N > 1 by default: the N = 1 workload (1M rows) split by rows gives `value`,
    then the metric's 10M-row base, split the same way, is measured by the same code and folded into `paths`."""
    shard = args.shard or ("rows" if be.world > 1 else "queries")
    if not (shard == "rows" and be.world > 1 and args.rows is None and args.big_rows > 0):
        return run_workload(args, be, args.rows if args.rows is not None else args.headline_rows, "only")
    head = run_workload(args, be, args.headline_rows, "headline")
    big = run_workload(args, be, args.big_rows, "big", outer=head)
    if head is not None and big is not None:
        cfg = big["config"]
        entry = {"path": "flat10m_sharded", "workload": cfg["workload"], "queries": cfg["queries"], "n_gpus": big["n_gpus"],
                 "ms_per_batch": big["ms_per_step"], "qps": big["value"], "first_pass": cfg["first_pass"],
                 "rows_total": cfg["rows_total"], "rows_per_gpu": cfg["rows_per_gpu"], "sharding": cfg["sharding"],
                 "exchange": cfg["exchange"], "roofline": big["roofline"], "parity": big["parity"],
                 "single_gpu_same_workload": cfg.get("single_gpu_same_workload"),
                 "refined_per_batch": big["refined_queries_per_step"], "fallback_per_batch": big["fallback_queries_per_step"],
                 "index_build_s": big["index_build_s"]}
        head["paths"] = [entry] + list(big.get("paths") or [])
    return head


def run_workload(args, be, rows_total, role, outer=None):
    """One flat workload of `rows_total` rows.  role "only": everything the line carries; "headline": the N > 1 headline (no
    IVF leg); "big": the larger base behind it (one first-pass precision, no CPU legs; `outer` is the headline's line, which
    the watchdog of the collective IVF leg prints should a rank stall)."""
    from needle_amd.distance import DistanceFunction
    from needle_amd.sharded import shard_bounds

    rank, world = be.rank, be.world
    shard = args.shard or ("rows" if world > 1 else "queries")
    by_rows = shard == "rows" and world > 1
    spans = [shard_bounds(rows_total, world, r, GEN_CHUNK) for r in range(world)] if by_rows else [(0, rows_total)] * world
    lo, hi = spans[rank]
    metric = DistanceFunction.from_str(args.metric)

    base = be.rows(hi - lo, args.dim, 42, first_row=lo)
    # query-sharded replicas draw different queries per rank; row shards all serve the same ones
    queries = be.rows(args.queries, args.dim, 42 + 999_999 + (0 if by_rows else rank))
    be.sync()

    t_build = time.perf_counter()
    index = be.flat_index(base, metric)
    t_build = time.perf_counter() - t_build
    search = {prec: be.searcher(index, prec, lo, by_rows) for prec in ("fp32", "bf16")}

    def timed(prec, steps, warmup):
        """(elapsed seconds for `steps` searches, library metrics over them, last result)."""
        out = None
        for _ in range(warmup):
            out = search[prec](queries, args.k)
        be.profile_begin()
        be.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = search[prec](queries, args.k)
        be.barrier()
        el = time.perf_counter() - t0
        mm = be.profile_end()
        return be.max_over_ranks(el), mm, out

    def roofline_of(prec, mm):
        # dominant kernel = first-pass scan: algorithmic flops = 2 * Q * rows_local * dim per launch, over the
        # launch time the library measures with HIP events on its own stream (seeding pre-pass included)
        scan_ms = mm["scan_ms_total"] / max(mm["scan_launches"], 1)
        flops = 2.0 * args.queries * (hi - lo) * args.dim
        achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if prec == "fp32" else BF16_MFMA_PEAK_TFLOPS
        kernel = "flat_scan_" + ("f32" if prec == "fp32" else "bf16")
        rec = counter_record(kernel)
        line = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "traffic": args.traffic_bytes if args.traffic_bytes is not None else (rec or {}).get("hbm_bytes_per_launch"),
                "launch_ms": round(scan_ms, 3), "launches_per_step": mm["scan_launches"] / max(mm["searches"], 1)}
        if rec and rec.get("held_clock_ghz"):
            # `peak` is priced at the 2.4 GHz boost clock; under this kernel the chip held `held_clock_ghz` (counter pass:
            # GRBM_GUI_ACTIVE / 8 / kernel time), so the MFMA pipes could have delivered peak * held / 2.4 at most
            line["held_clock_ghz"] = rec["held_clock_ghz"]
            line["frac_of_peak_at_held_clock"] = round(achieved / (peak * rec["held_clock_ghz"] / BOOST_CLOCK_GHZ), 4)
        return line

    def counter_record(kernel):
        """What a separate rocprofv3 --pmc pass over this same workload recorded for `kernel` (HBM bytes per search,
        the clock held), or None: when no pass was recorded, the workload differs, or the kernel's sources changed
        since (the record carries their hash)."""
        try:
            with open(os.path.join(ROOT, "needle_amd", "measured_traffic.json")) as fh:
                rec = json.load(fh).get(kernel)
        except (OSError, ValueError):
            return None
        want = {"rows": hi - lo, "dim": args.dim, "queries": args.queries, "k": args.k, "metric": args.metric}
        if not rec or any(rec.get("workload", {}).get(key) != val for key, val in want.items()):
            return None
        if rec.get("source_sha256") != kernel_source_hash(kernel):
            return None
        return rec

    elapsed, m, out = timed(args.precision, args.steps, args.warmup)
    ids, dists, counts = out
    ms_per_step = elapsed / args.steps * 1e3
    total_queries = args.queries * (1 if by_rows or world == 1 else world)
    qps = total_queries * args.steps / elapsed
    roofline = roofline_of(args.precision, m)

    secondary = None
    if args.secondary_steps > 0 and role != "big":
        other = "fp32" if args.precision == "bf16" else "bf16"
        el2, m2, _ = timed(other, args.secondary_steps, 1)
        secondary = {"first_pass": other, "value": round(total_queries * args.secondary_steps / el2, 1),
                     "unit": "queries/s", "ms_per_step": round(el2 / args.secondary_steps * 1e3, 3),
                     "roofline": roofline_of(other, m2)}

    cpu_baseline = parity = paths = None
    orc = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        orc = load_oracle()
        threads = min(orc.cpu_threads(), 16)
        hb, hq = be.host(base), be.host(queries)
        t1 = time.perf_counter()
        orc.flat_batch_search(hq[:threads], hb, args.k, int(metric), threads=threads)
        one = time.perf_counter() - t1
        rounds = int(max(1, min(16, args.cpu_seconds / max(one, 1e-3) - 1)))
        nq = max(1, min(args.queries - threads, rounds * threads))
        t1 = time.perf_counter()
        oi, od, oc = orc.flat_batch_search(hq[threads:threads + nq], hb, args.k, int(metric), threads=threads)
        dt = time.perf_counter() - t1
        cpu_baseline = {"value": round(nq / dt, 3), "unit": "queries/s", "cores": threads, "kind": "port",
                        "flags": oracle_build_flags(),
                        "sample": f"{nq} of the {args.queries} queries against the full {rows_total}x{args.dim} base, "
                                  f"{threads} threads, one query per task (oracle/needle_oracle.c, reference scalar order)"}
        gi = be.host(ids[threads:threads + nq]).view(np.uint64)
        gd = be.host(dists[threads:threads + nq])
        parity = {"queries_checked": int(nq), "ids_equal": bool(np.array_equal(gi, oi)),
                  "dists_bit_equal": bool(np.array_equal(gd.view(np.uint32), od.view(np.uint32)))}
        del hb
    if rank == 0 and world > 1 and args.parity_queries > 0:
        # N > 1: no CPU baseline (timed at N = 1 only), but rank 0 still checks a few answers against the oracle
        orc = load_oracle()
        nq = min(args.parity_queries, args.queries)
        hq = be.host(queries[:nq])
        oi, od, oc = oracle_sharded_truth(be, orc, args, metric, hq, spans if by_rows else [(0, rows_total)])
        gd = be.host(dists[:nq])
        parity = {"queries_checked": int(nq), "rank": 0,
                  "ids_equal": bool(np.array_equal(be.host(ids[:nq]).view(np.uint64), oi)),
                  "dists_bit_equal": bool(np.array_equal(gd.view(np.uint32), od.view(np.uint32)))}
    host_pointer = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and hasattr(be, "host_pointer_qps"):
        # the same batch through the host-pointer entry point: queries up and results down over PCIe (never `value`)
        host_pointer = be.host_pointer_qps(index, be.host(queries), args.k, args.precision)
    index.close()
    want_paths = [] if args.paths in ("", "none") else args.paths.split(",")
    sharded_ivf = by_rows and world > 1 and "ivf" in want_paths and hasattr(be, "ivf_sharded") and role != "headline"
    if not sharded_ivf:
        del base
    if rank == 0 and world == 1 and want_paths:
        be.hnsw_rows, be.hnsw_budget_s = args.hnsw_rows, args.hnsw_seconds
        paths = be.paths(want_paths, orc)

    line = None
    if rank == 0:
        sharding = (f"row-blocks x{world}, all-gather + merge" if by_rows
                    else (f"query-sharded replicas x{world}, no exchange" if world > 1 else "single GPU"))
        line = {
            "metric": METRIC_LABEL,
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong" if by_rows else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"exact flat batch_search, {rows_total}x{args.dim} {args.metric}, "
                                   f"{args.queries} queries, k={args.k}, unit-normalised Gaussian embeddings",
                       "rows_total": rows_total, "rows_per_gpu": hi - lo, "dim": args.dim, "queries": args.queries,
                       "k": args.k, "first_pass": args.precision, "queries_total": total_queries,
                       "sharding": sharding, "exchange": be.exchange_label() if by_rows else None,
                       "scaling_reference": ("`value` at every N is this same workload (the N = 1 run of this script); rank 0 also "
                                             "times it alone on its GPU in this run: single_gpu_same_workload")
                       if by_rows and world > 1 else None,
                       "recall_at_10": 1.0 if parity is not None and parity["ids_equal"] else None},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
            "other_first_pass": secondary, "host_pointer_qps": host_pointer,
            "refined_queries_per_step": m["refined_queries"] / max(args.steps, 1),
            "fallback_queries_per_step": m["fallback_queries"] / max(args.steps, 1),
            "index_build_s": round(t_build, 3),
            "paths": paths,
        }
    if sharded_ivf:
        # The flat line above is complete.  The IVF leg is collective: should a rank fail or stall in it, the others
        # would wait for ever, so a watchdog prints the flat line and ends the process instead of losing the run.
        import threading

        done = threading.Event()

        def bail():
            if done.is_set():
                return
            shown = outer if outer is not None else line  # the headline's line when this is the larger base behind it
            if shown is not None:
                shown["paths"] = [{"path": "ivf_sharded", "error": f"not finished within {args.ivf_leg_seconds:.0f} s"}]
                print(json.dumps(shown), flush=True)
            # a stalled collective leg is a failed run: the flat line is out, the exit status says so (3 = parity, 4 = stall)
            os._exit(4 if shown is None or parity_ok(shown) else 3)

        dog = threading.Timer(args.ivf_leg_seconds, bail)
        dog.daemon = True
        dog.start()
        try:
            leg = be.ivf_sharded(base, queries, lo, rows_total, args.k, ids, max(2, min(args.steps, 5)))
        except Exception as e:  # noqa: BLE001
            leg = {"path": "ivf_sharded", "error": f"{type(e).__name__}: {e}"}
        done.set()
        dog.cancel()
        del base
        if line is not None:
            line["paths"] = [leg]
    if by_rows and world > 1 and not args.no_scaling_reference:
        # The strong-scaling reference of THIS workload inside this run: rank 0 searches the whole base on its own GPU
        # while the others wait.
        if rank == 0:
            try:
                full = be.rows(rows_total, args.dim, 42)
                whole = be.flat_index(full, metric)
                one = be.searcher(whole, args.precision, 0, False)
                one(queries, args.k)
                be.sync()
                reps = max(2, min(args.steps, 5))
                t0 = time.perf_counter()
                for _ in range(reps):
                    one(queries, args.k)
                be.sync()
                dt = (time.perf_counter() - t0) / reps
                ref = {"n_gpus": 1, "value": round(args.queries / dt, 1), "unit": "queries/s", "ms_per_step": round(dt * 1e3, 3),
                       "workload": f"the same {rows_total}x{args.dim} base and query batch on one GPU, measured by rank 0 in this run"}
                whole.close()
                del full
            except Exception as e:  # noqa: BLE001
                ref = {"error": f"{type(e).__name__}: {e}"}
            line["config"]["single_gpu_same_workload"] = ref
        be.barrier()
    return line


def parity_ok(line):
    """False when any parity sample of the line (headline or paths) disagrees with the oracle."""
    checks = [line.get("parity")] + [p.get("parity") for p in (line.get("paths") or []) if isinstance(p, dict)]
    for c in checks:
        if isinstance(c, dict) and not all(v for key, v in c.items() if key.endswith("_equal")):
            return False
    return True


def visible_gpus():
    """GPUs this process could use, counted without initialising the runtime (the launcher must stay GPU-free)."""
    try:
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


def launch_ranks(args, argv):
    """`bench.py --gpus N` started without a launcher: start the N ranks ourselves, one process per GPU, with the
    environment torch.distributed.run would give them, relay their output and fail when any of them fails.  Runs before
    anything in this process touches a GPU; the children are ordinary child processes (never an exec)."""
    import socket
    import subprocess

    n = args.gpus
    transport = os.environ.get("NEEDLE_BENCH_BACKEND", "nccl")
    if args.backend == "hip":
        have = visible_gpus()
        if have == 0 or (transport == "nccl" and have < n):
            print(f"bench.py: --gpus {n} needs {n} visible GPUs for the RCCL exchange, found {have} "
                  f"(NEEDLE_BENCH_BACKEND=gloo rehearses the control flow with ranks sharing devices)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-u", os.path.abspath(__file__)] + list(argv if argv is not None else sys.argv[1:])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} ended with status {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:  # exactly the processes started above
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def make_backend(spec, rank, local_rank, world):
    if spec == "hip":
        return HipBackend(rank, local_rank, world)
    import importlib

    mod, _, attr = spec.partition(":")
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("gloo")  # stand-in backends exchange over host tensors
    return getattr(importlib.import_module(mod), attr)(rank, local_rank, world)


def main(argv=None, backend=None):
    args = parse(argv)
    if backend is None and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE=1: refusing to report one rank as {args.gpus} GPUs",
                  file=sys.stderr)
            return 2
        args.gpus = world
    be = backend(rank, local_rank, world) if backend is not None else make_backend(args.backend, rank, local_rank, world)
    try:
        line = run(args, be)
    finally:
        be.close()
        if args.backend != "hip" and dist.is_initialized():
            dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)
        if not parity_ok(line):
            print("bench.py: a parity sample disagrees with the oracle", file=sys.stderr)
            return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
