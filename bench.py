#!/usr/bin/env python3
"""Batch-search throughput of the exact flat scan on MI355X.

A "step" is one complete batch search: Q queries against the resident base, ids and distances of the
k nearest per query delivered (first pass on the matrix cores, candidate merge, reference-order
re-scoring, bound check, exact scan for rejected queries; with more than one GPU also the all-gather
of per-shard results and the merge).  Inputs are resident in HBM before the timed region starts; the
index build (upload, norms, operand copy) is outside it, like the reference's insert phase.

Default workload (N=1): exact flat batch_search, 1M x 384 cosine, 10K queries, k=10, synthetic
unit-normalised Gaussian embeddings, bf16 first pass (the delivered ids and distances are those of
the reference's f32 scalar arithmetic whatever the first pass; the fp32 first pass is timed beside it).

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU):
  --shard queries (default): every GPU holds the whole base (1M x 384 is 1.5 GB of 288 GB) and serves
      its own 10K queries; no exchange on the data path; per-GPU work is fixed -> weak scaling.
  --shard rows: the base is split into N contiguous row blocks, every GPU scans its block for ALL
      queries, per-shard results are all-gathered over RCCL and merged -> strong scaling of one batch;
      the layout for bases beyond one GPU's memory.
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
GEN_CHUNK = 5000                # rows per generator stream; shard starts are multiples of it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1_000_000, help="total base rows (split across GPUs)")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--shard", default="queries", choices=["queries", "rows"])
    ap.add_argument("--secondary-steps", type=int, default=2, help="steps of the other first-pass precision (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per first-pass launch from a separate rocprofv3 --pmc run")
    return ap.parse_args()


def synth(n, dim, seed, device, first_row=0):
    """Unit-normalised Gaussian rows; row i depends only on (seed, i), so shards agree with the whole."""
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    chunk = GEN_CHUNK
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + (first_row + s) // chunk)
        blk = torch.randn((chunk, dim), generator=g, device=device, dtype=torch.float32)
        blk = blk / blk.norm(dim=1, keepdim=True)
        off = (first_row + s) % chunk
        assert off == 0, "shard boundaries must fall on generator chunks"
        out[s:e] = blk[: e - s]
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # NEEDLE_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks (ranks
    # share devices, timing reduced over CPU tensors); the measured runs use RCCL, one rank per GPU.
    backend = os.environ.get("NEEDLE_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import needle_amd as na
    from needle_amd.sharded import ShardedSearcher, hip_local_search, hip_merge

    chunk = GEN_CHUNK
    by_rows = args.shard == "rows" and world > 1
    if by_rows:
        per = -(-args.rows // world)
        per = -(-per // chunk) * chunk  # keep shard starts on generator chunks
        lo = min(rank * per, args.rows)
        hi = min(lo + per, args.rows)
    else:
        lo, hi = 0, args.rows
    metric = na.DistanceFunction.from_str(args.metric)

    base = synth(hi - lo, args.dim, 42, device, first_row=lo)
    # query-sharded replicas draw different queries per rank; row shards all serve the same ones
    queries = synth(args.queries, args.dim, 42 + 999_999 + (0 if by_rows else rank), device)
    torch.cuda.synchronize()

    ctx = na.Context(dev_index)
    t_build = time.perf_counter()
    index = na.FlatIndex(ctx, base, metric)
    t_build = time.perf_counter() - t_build
    local = {prec: hip_local_search(index, prec) for prec in ("fp32", "bf16")}
    sharded = {prec: ShardedSearcher(local[prec], hip_merge(ctx), lo) for prec in local} if by_rows else None

    def search(prec):
        return sharded[prec].search(queries, args.k) if by_rows else local[prec](queries, args.k)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(prec, steps, warmup):
        """(elapsed seconds for `steps` searches, library metrics over them, last result)."""
        out = None
        for _ in range(warmup):
            out = search(prec)
        ctx.set_profiling(True)
        ctx.reset_metrics()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = search(prec)
        barrier()
        el = time.perf_counter() - t0
        mm = ctx.metrics()
        ctx.set_profiling(False)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, mm, out

    def roofline_of(prec, mm):
        # dominant kernel = first-pass scan: algorithmic flops = 2 * Q * rows_local * dim per launch, over the
        # launch time the library measures with HIP events on its own stream (seeding pre-pass included)
        scan_ms = mm["scan_ms_total"] / max(mm["scan_launches"], 1)
        flops = 2.0 * args.queries * (hi - lo) * args.dim
        achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if prec == "fp32" else BF16_MFMA_PEAK_TFLOPS
        kernel = "flat_scan_" + ("f32" if prec == "fp32" else "bf16")
        return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": measured_traffic(kernel), "launch_ms": round(scan_ms, 3),
                "launches_per_step": mm["scan_launches"] / max(mm["searches"], 1)}

    def measured_traffic(kernel):
        """HBM bytes per launch from a separate rocprofv3 --pmc pass over this same workload, if recorded."""
        if args.traffic_bytes is not None:
            return args.traffic_bytes
        try:
            with open(os.path.join(ROOT, "needle_amd", "measured_traffic.json")) as fh:
                rec = json.load(fh).get(kernel)
        except (OSError, ValueError):
            return None
        want = {"rows": hi - lo, "dim": args.dim, "queries": args.queries, "k": args.k, "metric": args.metric}
        if not rec or any(rec.get("workload", {}).get(key) != val for key, val in want.items()):
            return None
        return rec["hbm_bytes_per_launch"]

    elapsed, m, out = timed(args.precision, args.steps, args.warmup)
    ids, dists, counts = out
    ms_per_step = elapsed / args.steps * 1e3
    total_queries = args.queries * (1 if by_rows else world)
    qps = total_queries * args.steps / elapsed
    roofline = roofline_of(args.precision, m)

    secondary = None
    if args.secondary_steps > 0:
        other = "fp32" if args.precision == "bf16" else "bf16"
        el2, m2, _ = timed(other, args.secondary_steps, 1)
        secondary = {"first_pass": other, "value": round(total_queries * args.secondary_steps / el2, 1),
                     "unit": "queries/s", "ms_per_step": round(el2 / args.secondary_steps * 1e3, 3),
                     "roofline": roofline_of(other, m2)}

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import oracle as orc  # checker + CPU baseline only; never on the product path

        orc.build()
        threads = min(orc.cpu_threads(), 16)
        hb = base.cpu().numpy()
        hq = queries.cpu().numpy()
        t1 = time.perf_counter()
        orc.flat_batch_search(hq[:threads], hb, args.k, int(metric), threads=threads)
        one = time.perf_counter() - t1
        rounds = int(max(1, min(16, args.cpu_seconds / max(one, 1e-3) - 1)))
        nq = min(args.queries - threads, rounds * threads)
        t1 = time.perf_counter()
        oi, od, oc = orc.flat_batch_search(hq[threads:threads + nq], hb, args.k, int(metric), threads=threads)
        dt = time.perf_counter() - t1
        cpu_baseline = {"value": round(nq / dt, 3), "unit": "queries/s", "cores": threads, "kind": "port",
                        "sample": f"{nq} of the {args.queries} queries against the full {args.rows}x{args.dim} base, "
                                  f"{threads} threads, one query per task (oracle/needle_oracle.c, reference scalar order)"}
        gi = ids[threads:threads + nq].cpu().numpy().view(np.uint64)
        gd = dists[threads:threads + nq].cpu().numpy()
        parity = {"queries_checked": int(nq), "ids_equal": bool(np.array_equal(gi, oi)),
                  "dists_bit_equal": bool(np.array_equal(gd.view(np.uint32), od.view(np.uint32)))}

    if rank == 0 and world > 1 and not by_rows and args.cpu_seconds > 0:
        # N > 1: no CPU baseline (timed at N = 1 only), but rank 0 still checks a few of its answers
        from oracle import oracle as orc  # checker only

        orc.build()
        nq = min(32, args.queries)
        oi, od, oc = orc.flat_batch_search(queries[:nq].cpu().numpy(), base.cpu().numpy(), args.k, int(metric),
                                           threads=min(orc.cpu_threads(), 16))
        gd = dists[:nq].cpu().numpy()
        parity = {"queries_checked": int(nq), "rank": 0,
                  "ids_equal": bool(np.array_equal(ids[:nq].cpu().numpy().view(np.uint64), oi)),
                  "dists_bit_equal": bool(np.array_equal(gd.view(np.uint32), od.view(np.uint32)))}

    if rank == 0:
        line = {
            "metric": "batch-search QPS @ recall@10, 1M & 10M\u00d7384 cosine k=10, 1/2/4/8 MI355X",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong" if by_rows else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"exact flat batch_search, {args.rows}x{args.dim} {args.metric}, "
                                   f"{args.queries} queries, k={args.k}, unit-normalised Gaussian embeddings",
                       "rows_total": args.rows, "rows_per_gpu": hi - lo, "dim": args.dim, "queries": args.queries,
                       "k": args.k, "first_pass": args.precision, "queries_total": total_queries,
                       "sharding": (f"row-blocks x{world}, all-gather + merge" if by_rows
                                    else f"query-sharded replicas x{world}, no exchange"),
                       "recall_at_10": 1.0 if parity is not None and parity["ids_equal"] else None},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
            "other_first_pass": secondary,
            "refined_queries_per_step": m["refined_queries"] / max(args.steps, 1),
            "fallback_queries_per_step": m["fallback_queries"] / max(args.steps, 1),
            "index_build_s": round(t_build, 3),
        }
        print(json.dumps(line))
    index.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
