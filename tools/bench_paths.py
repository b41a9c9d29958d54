#!/usr/bin/env python3
"""Throughput of the paths beside the headline flat scan, each on ONE MI355X at the per-GPU share of
the configurations BASELINE.json lists (one JSON line per path):

  c1   exact flat, 100K x 128 L2, 1K queries, k=10                      (plumbing shape)
  c5   exact flat, 1M x 768 dot product, 10 % filter bitmask, k=100, 1K queries
  ivf  IVF-Flat nlist=4096 nprobe=32 over 1.25M x 384 (= 10M / 8 GPUs), 10K queries, k=10
  ivf_clustered  the same over a mixture of 4096 Gaussians (recall@10 is meaningful there)
  sq8  SQ8 asymmetric scan over 6.25M x 384 codes (= 50M / 8), 1K queries, k=10
  pq   PQ m=48 table scan over 6.25M x 48 codes (= 50M / 8), 1K queries, k=10
  binary  1-bit codes, Hamming scan over 6.25M x 384 bits (= 50M / 8), 1K queries, k=10
  hnsw (opt-in: CPU minutes)  what the reference's Collection::batch_search runs today -- its HNSW index at the default
       configuration, restated on the CPU -- on a bounded sample of the headline workload, with recall@10 against the
       library's exact result over the same rows and the library's own time beside it

Inputs are synthetic (Gaussian rows; IVF centroids = k-means++ and 8 Lloyd iterations on a 100-rows-per-list sample, trained on the device, quantizer
parameters trained on a sample; codes of the big sets are drawn uniformly, which is what a scan's
cost depends on).  Queries and results live in HBM (device-pointer entry point).  A sample of each
result is checked against the CPU oracle where the oracle finishes in seconds.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import needle_amd as na  # noqa: E402
import needle_amd.ivf  # noqa: E402,F401
import needle_amd.quantization  # noqa: E402,F401
from needle_amd import _lib  # noqa: E402
from needle_amd.index import PRECISION, _mask_words  # noqa: E402

F = np.float32
HBM_PEAK_GBS = 8000.0
BF16_PEAK_TFLOPS = 2500.0
FP32_PEAK_TFLOPS = 157.3


def search_device(handle, dq, k, *, nprobe=0, precision="fp32", d_filter=None):
    nq = dq.shape[0]
    ids = torch.empty((nq, k), dtype=torch.int64, device=dq.device)
    dists = torch.empty((nq, k), dtype=torch.float32, device=dq.device)
    counts = torch.empty((nq,), dtype=torch.int32, device=dq.device)
    sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0,
                           filter_bitmask=d_filter.data_ptr() if d_filter is not None else None)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().needle_hip_batch_search_device(handle, dq.data_ptr(), nq, C.byref(sp), ids.data_ptr(),
                                                         dists.data_ptr(), counts.data_ptr()))
    return ids, dists, counts


def timed(fn, steps, warmup):
    """Seconds per call over `steps` calls.  At least two untimed calls come first: an index adapts the number of candidates
    it keeps after its first batch (and grows its workspace once more), which belongs to start-up, not to the steady rate."""
    out = None
    for _ in range(max(warmup, 2)):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def concurrent_qps(handle, dq, k, callers, steps, *, nprobe=0, precision="bf16", d_filter=None):
    """Aggregate rate of `callers` host threads searching ONE index at the same time (the library runs every call on a
    search slot of its own: private workspace and stream), each with its own result buffers."""
    import threading

    nq = dq.shape[0]
    sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0,
                           filter_bitmask=d_filter.data_ptr() if d_filter is not None else None)
    bufs = [(torch.empty((nq, k), dtype=torch.int64, device=dq.device), torch.empty((nq, k), dtype=torch.float32, device=dq.device),
             torch.empty((nq,), dtype=torch.int32, device=dq.device)) for _ in range(callers)]
    torch.cuda.synchronize()
    gate = threading.Barrier(callers + 1)
    errs = []

    def work(t):
        ids, dists, counts = bufs[t]
        try:
            for r in range(steps + 1):
                if r == 1:
                    gate.wait()  # one untimed call each (slot creation, workspace), then everybody starts together
                _lib.check(_lib.lib().needle_hip_batch_search_device(handle, dq.data_ptr(), nq, C.byref(sp), ids.data_ptr(),
                                                                     dists.data_ptr(), counts.data_ptr()))
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            gate.abort()

    ths = [threading.Thread(target=work, args=(t,)) for t in range(callers)]
    for th in ths:
        th.start()
    gate.wait()
    t0 = time.perf_counter()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    return round(callers * steps * nq / dt, 1)


CONCURRENCY = False  # --concurrency: also time 2 and 4 host threads searching one index at once

COLLECT = None  # a list: entries are appended instead of printed (bench.py folds them into its JSON line)


def emit(name, workload, nq, sec, extra):
    line = {"path": name, "workload": workload, "queries": nq, "ms_per_batch": round(sec * 1e3, 3),
            "qps": round(nq / sec, 1)}
    line.update(extra)
    if COLLECT is not None:
        COLLECT.append(line)
    else:
        print(json.dumps(line), flush=True)


def roofline(bound, achieved, unit, kernel, per_unit, launch_ms):
    """The roofline object of one path: `achieved` = algorithmic bytes or flops of one batch over `launch_ms`."""
    peak = HBM_PEAK_GBS if bound == "hbm" else (BF16_PEAK_TFLOPS if "bf16" in kernel or "mfma" in kernel else FP32_PEAK_TFLOPS)
    return {"bound": bound, "kernel": kernel, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "algorithmic": per_unit, "launch_ms": round(launch_ms, 3), "traffic": None}


def host_api_qps(handle, queries_host, k, *, nprobe=0, precision="bf16", filt_words=None, reps=2):
    """The same batch through the host-pointer entry point (pageable numpy buffers: queries up, results down over PCIe)."""
    q = np.ascontiguousarray(queries_host, dtype=F)
    nq, dim = q.shape
    ids = np.empty((nq, k), dtype=np.uint64)
    dists = np.empty((nq, k), dtype=F)
    counts = np.empty(nq, dtype=np.uint32)
    sp = _lib.SearchParams(k=k, nprobe=nprobe, precision=PRECISION[precision], reserved=0,
                           filter_bitmask=filt_words.ctypes.data if filt_words is not None else None)
    best = None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        _lib.check(_lib.lib().needle_hip_batch_search(handle, q.ctypes.data, nq, dim, C.byref(sp), ids.ctypes.data,
                                                      dists.ctypes.data, counts.ctypes.data))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return round(nq / best, 1)


def bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint32)


def flat_case(ctx, dev, name, n, dim, nq, k, metric, steps, filt_frac=None, oracle=None, normalise=False, precisions=("bf16", "fp32"),
              parity_queries=32):
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    base = torch.empty((n, dim), device=dev, dtype=torch.float32)
    for s0 in range(0, n, 1 << 20):  # in pieces: randn's temporaries stay small beside a 15 GB base
        base[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), dim), generator=g, device=dev, dtype=torch.float32)
    queries = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
    if normalise:  # "normalized embeddings": the reference's normalize recipe, on the device
        torch.cuda.synchronize()
        ctx.normalize_device(base)
        ctx.normalize_device(queries)
    ix = na.FlatIndex(ctx, base, metric)
    d_filter = keep = None
    if filt_frac:
        keep = (np.random.default_rng(7).random(n) < filt_frac)
        d_filter = torch.from_numpy(_mask_words(keep, n).view(np.int32)).to(dev)
    for prec in precisions:
        search_device(ix._h, queries, k, precision=prec, d_filter=d_filter)  # workspaces are allocated outside the timed events
        ctx.set_profiling(True)
        ctx.reset_metrics()
        sec, out = timed(lambda: search_device(ix._h, queries, k, precision=prec, d_filter=d_filter), steps, 1)
        m = ctx.metrics()
        ctx.set_profiling(False)
        scan_ms = m["scan_ms_total"] / max(m["scan_launches"], 1)
        # A filtered search that admits a small share of the rows scans a gathered copy of them: the rows it must score
        # are the admitted ones, and the roofline counts those (the gather itself is in ms_per_batch, not in scan_ms).
        compacted = m["compacted_searches"] > 0
        rows_scored = m["compacted_rows"] // m["compacted_searches"] if compacted else n
        flops = 2.0 * nq * rows_scored * dim
        peak = 2500.0 if prec == "bf16" else 157.3
        extra = {"first_pass": prec, "scan_ms": round(scan_ms, 3), "scan_tflops": round(flops / scan_ms / 1e9, 1),
                 "frac_of_mfma_peak": round(flops / scan_ms / 1e9 / peak, 3),
                 "roofline": roofline("mfma", flops / scan_ms / 1e9, "TFLOP/s", "flat_scan_" + ("bf16" if prec == "bf16" else "f32"),
                                      "2*Q*rows_scored*D flop per batch", scan_ms),
                 "refined_per_batch": m["refined_queries"] / (steps + 1), "fallback_per_batch": m["fallback_queries"] / (steps + 1)}
        if filt_frac:
            extra.update({"rows_scored": int(rows_scored), "admitted_rows_gathered": bool(compacted),
                          "whole_base_equivalent_tflops": round(2.0 * nq * n * dim / (sec * 1e3) / 1e9, 1),
                          "gather_bytes": int(rows_scored) * dim * (4 + (2 if prec == "bf16" else 0)) * 2 if compacted else 0})
        if prec == precisions[0]:
            extra["host_pointer_qps"] = host_api_qps(ix._h, queries.cpu().numpy(), k, precision=prec,
                                                     filt_words=_mask_words(keep, n) if keep is not None else None)
            if nq <= 2000 and CONCURRENCY:  # small batches leave the chip idle between their short kernels: what concurrent callers gain
                extra["concurrent_callers_qps"] = {str(t): concurrent_qps(ix._h, queries, k, t, max(steps, 4), precision=prec,
                                                                          d_filter=d_filter) for t in (2, 4)}
        if oracle is not None:
            s = min(nq, parity_queries)
            oi, od, oc = oracle.flat_batch_search(queries[:s].cpu().numpy(), base.cpu().numpy(), k, int(metric), filt=keep,
                                                  threads=min(oracle.cpu_threads(), 16))
            gi = out[0][:s].cpu().numpy().view(np.uint64)
            extra["parity"] = {"queries_checked": s, "ids_equal": bool(np.array_equal(gi, oi)),
                               "dists_bit_equal": bool(np.array_equal(bits(out[1][:s].cpu().numpy()), bits(od)))}
        emit(name, f"exact flat {n}x{dim} {metric.name}, k={k}" + (f", {int(filt_frac * 100)} % filter" if filt_frac else ""),
             nq, sec, extra)
    ix.close()


def trained_centroids(ctx, sample, nlist, iters, seed=11):
    """IvfIndex::train on the device (needle_hip_ivf_train): k-means++ and `iters` Lloyd iterations."""
    ix = na.ivf.IvfIndex(sample.shape[1], na.ivf.IvfConfig.new(nlist).with_max_iterations(iters), ctx=ctx, seed=seed)
    t0 = time.perf_counter()
    ix.train(sample)
    return ix.centroids, time.perf_counter() - t0


def ivf_case(ctx, dev, n, dim, nlist, nprobe, nq, k, steps, oracle=None, centroids="trained", normalise=True, data="gaussian"):
    """data = "gaussian": i.i.d. rows (no cluster structure: recall@10 of ANY inverted file is then a few percent);
    "clustered": a mixture of `nlist` Gaussians (centres N(0, 1), spread 0.5 per component), the structure embeddings
    have -- recall@10 against the exact search says something there.  Same seeds in both modes."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((n, dim), dtype=F)
    queries = rng.standard_normal((nq, dim), dtype=F)
    if data == "clustered":
        crng = np.random.default_rng(12)
        centres = crng.standard_normal((nlist, dim), dtype=F)
        base *= F(0.5)
        queries *= F(0.5)
        for s0 in range(0, n, 1 << 18):  # in pieces: the gathered centres of 1.25M rows are 1.9 GB
            base[s0:s0 + (1 << 18)] += centres[crng.integers(0, nlist, min(1 << 18, n - s0))]
        queries += centres[crng.integers(0, nlist, nq)]
    if normalise:  # the configuration's "cosine": IvfIndex ranks by Euclidean distance (ivf.rs:67), over unit rows the same order
        base = ctx.normalize(base)
        queries = ctx.normalize(queries)
    cents, train_s = np.ascontiguousarray(base[rng.choice(n, nlist, replace=False)]), 0.0
    if centroids == "trained":  # k-means on a sample of 100 rows per list, as an index is trained
        cents, train_s = trained_centroids(ctx, base[rng.choice(n, min(n, 100 * nlist), replace=False)], nlist, 8)
    h = C.c_void_p()
    t0 = time.perf_counter()
    _lib.check(_lib.lib().needle_hip_ivf_build(ctx._h, base.ctypes.data, n, dim, cents.ctypes.data, nlist, None, None,
                                               C.byref(h)))
    build_s = time.perf_counter() - t0
    dq = torch.from_numpy(queries).to(dev)
    sec32, _ = timed(lambda: search_device(h, dq, k, nprobe=nprobe, precision="fp32"), steps, 1)
    ctx.set_profiling(True)
    ctx.reset_metrics()
    sec, out = timed(lambda: search_device(h, dq, k, nprobe=nprobe, precision="bf16"), steps, 1)
    mm = ctx.metrics()
    ctx.set_profiling(False)
    list_scan_ms = mm["list_scan_ms_total"] / max(mm["list_scans"], 1)
    sizes = np.zeros(nlist, dtype=np.uint32)
    _lib.check(_lib.lib().needle_hip_ivf_list_sizes(h, sizes.ctypes.data))
    probe = np.full((nq, nprobe), 0xFFFFFFFF, dtype=np.uint32)
    pc = np.zeros(nq, dtype=np.uint32)
    _lib.check(_lib.lib().needle_hip_ivf_probe(h, queries.ctypes.data, nq, dim, nprobe, probe.ctypes.data, pc.ctypes.data))
    per_list = np.bincount(probe[probe != 0xFFFFFFFF].astype(np.int64), minlength=nlist)
    padded = (sizes.astype(np.int64) + 127) // 128 * 128
    row_bytes = dim * 4
    bytes_items = int((((per_list + 127) // 128) * padded).sum()) * row_bytes      # what the list-major scan reads
    bytes_once = int(padded[per_list > 0].sum()) * row_bytes                         # every probed list once
    bytes_unbatched = int((per_list * sizes.astype(np.int64)).sum()) * row_bytes     # per-query scans, as the reference
    bytes_once_rows = int(sizes.astype(np.int64)[per_list > 0].sum()) * row_bytes   # every probed list once, its rows only: <= 4*N*D
    pairs = int((per_list * sizes.astype(np.int64)).sum())
    # recall@10 against the exact flat search of the same rows (the library's own, checked against the oracle elsewhere)
    flat = na.FlatIndex(ctx, base, na.DistanceFunction.Euclidean)
    exact = search_device(flat._h, dq, k, precision="bf16")[0].cpu().numpy()
    flat.close()
    got_ids = out[0].cpu().numpy()

    def recall_of(found, truth):
        return float(np.mean([len(np.intersect1d(found[i], truth[i])) / max(1, truth.shape[1]) for i in range(found.shape[0])]))

    recall_all = recall_of(got_ids, exact)
    bytes_bf16 = bytes_items // 2  # what the bf16 list scan streams (2 bytes per value)
    rows_label = ("unit-normalised " if normalise else "") + ("mixture of %d Gaussians" % nlist if data == "clustered" else "Gaussian")
    extra = {"centroids": centroids, "rows": rows_label, "largest_list": int(sizes.max()), "list_scan": "bf16",
             "recall_at_10": round(recall_all, 4),
             "list_scan_ms": round(list_scan_ms, 3),
             "list_scan_only_GBps_f32_rows": round(bytes_items / list_scan_ms / 1e6, 1) if list_scan_ms > 0 else None,
             "list_scan_only_GBps_bf16_streamed": round(bytes_bf16 / list_scan_ms / 1e6, 1) if list_scan_ms > 0 else None,
             "roofline": roofline("hbm", bytes_items / sec / 1e9, "GB/s", "ivf list-major scan (whole search)",
                                  "4*D bytes per row of every probed list, once per tile of the queries probing it", sec * 1e3),
             "host_pointer_qps": host_api_qps(h, queries, k, nprobe=nprobe, precision="bf16"),
             "concurrent_callers_qps": ({str(t): concurrent_qps(h, dq, k, t, max(steps, 4), nprobe=nprobe) for t in (2, 4)}
                                        if CONCURRENCY else None),
             "fp32_list_scan_ms_per_batch": round(sec32 * 1e3, 3), "build_s": round(build_s, 2), "train_s": round(train_s, 2),
             "scanned_pairs": pairs,
             "bytes_listmajor": bytes_items, "bytes_each_probed_list_once": bytes_once_rows,
             "bytes_each_probed_list_once_with_tile_padding": bytes_once, "bytes_unbatched": bytes_unbatched,
             "frac_of_hbm_peak_each_probed_list_once": round(bytes_once_rows / sec / 1e9 / HBM_PEAK_GBS, 4),
             "listmajor_GBps_over_whole_search": round(bytes_items / sec / 1e9, 1),
             "frac_of_hbm_peak": round(bytes_items / sec / 1e9 / HBM_PEAK_GBS, 4),
             "pair_gflops_over_whole_search": round(2.0 * pairs * dim / sec / 1e9, 1)}
    if oracle is not None:
        s = 24
        # The oracle's own assignment pass is n x nlist x dim scalar MACs (minutes at this size).  Beyond a small
        # size the lists the oracle scans are built from the DEVICE assignment instead, which tests/test_ivf_gpu.py
        # checks bit for bit against the oracle's; probe lists and the search itself are always the oracle's.
        small = float(n) * nlist * dim < 3e11
        assign = oracle.ivf_assign(base, cents) if small else ctx.kmeans_assign(base, cents, 0)
        offsets, order = oracle.ivf_lists_from_assignment(assign, nlist)
        oi, od, oc = oracle.ivf_batch_search(queries[:s], base, cents, offsets, order, nprobe, k)
        gi = out[0][:s].cpu().numpy().view(np.uint64)
        op = np.stack([oracle.ivf_probe(queries[i], cents, nprobe) for i in range(s)])
        extra["parity"] = {"queries_checked": s, "assignment_from": "oracle" if small else "device (tested against the oracle)",
                           "ids_equal": bool(np.array_equal(gi, oi)),
                           "dists_bit_equal": bool(np.array_equal(bits(out[1][:s].cpu().numpy()), bits(od))),
                           "probe_lists_equal": bool(np.array_equal(probe[:s].astype(np.int64), op.astype(np.int64))),
                           # the same queries' recall from the oracle's inverted file and from the library's: equal by construction
                           "recall_at_10_oracle_ivf": round(recall_of(oi.astype(np.int64), exact[:s]), 4),
                           "recall_at_10_library_ivf": round(recall_of(got_ids[:s], exact[:s]), 4)}
    emit("ivf" if data == "gaussian" else "ivf_" + data, f"IVF-Flat nlist={nlist} nprobe={nprobe} over {n}x{dim} Euclidean ({rows_label} rows), k={k}",
         nq, sec, extra)
    _lib.lib().needle_hip_index_destroy(h)


def sq8_case(ctx, dev, n, dim, nq, k, steps, oracle=None):
    rng = np.random.default_rng(13)
    sample = rng.standard_normal((20000, dim), dtype=F)
    sq = na.quantization.ScalarQuantizer.train(sample)
    codes = rng.integers(0, 256, size=(n, dim), dtype=np.uint8)
    codes[:20000] = sq.quantize(sample)
    queries = rng.standard_normal((nq, dim), dtype=F)
    ix = na.quantization.Sq8Index(ctx, sq, codes)
    dq = torch.from_numpy(queries).to(dev)
    for prec in ("bf16", "fp32"):
        ctx.set_profiling(True)
        ctx.reset_metrics()
        sec, out = timed(lambda: search_device(ix._h, dq, k, precision=prec), steps, 1)
        m = ctx.metrics()
        ctx.set_profiling(False)
        scan_ms = m["scan_ms_total"] / max(m["scan_launches"], 1)
        # every query block streams the operand matrix once: 256 queries x 2-byte operands, or 128 x 1-byte codes
        byts = (-(-nq // 256)) * n * dim * 2 if prec == "bf16" else (-(-nq // 128)) * n * dim
        algo = (-(-nq // (256 if prec == "bf16" else 128))) * n * dim  # D bytes per vector, once per query block
        extra = {"first_pass": prec, "scan_ms": round(scan_ms, 3), "bytes_streamed": byts,
                 "streamed_GBps": round(byts / scan_ms / 1e6, 1),
                 "roofline": roofline("hbm", algo / scan_ms / 1e6, "GB/s", "sq8 scan, " + prec + " operands",
                                      "D bytes per vector, once per query block", scan_ms),
                 "scan_tflops": round(2.0 * nq * n * dim / scan_ms / 1e9, 1),
                 # what actually binds this scan: the codes are multiplied on the matrix cores (every u8 is exact in bf16)
                 "roofline_mfma": roofline("mfma", 2.0 * nq * n * dim / scan_ms / 1e9, "TFLOP/s", "flat_scan_" + ("bf16" if prec == "bf16" else "f32"),
                                           "2*Q*N*D flop per batch", scan_ms),
                 "refined_per_batch": m["refined_queries"] / (steps + 1),
                 "fallback_per_batch": m["fallback_queries"] / (steps + 1)}
        if prec == "bf16":
            extra["host_pointer_qps"] = host_api_qps(ix._h, queries, k, precision=prec)
            if CONCURRENCY:
                extra["concurrent_callers_qps"] = {str(t): concurrent_qps(ix._h, dq, k, t, max(steps, 4), precision=prec) for t in (2, 4)}
        if oracle is not None:
            s, ns = 8, 400_000
            small = na.quantization.Sq8Index(ctx, sq, codes[:ns])
            gi, gd, gc = small.search(queries[:s], k, precision=prec)
            osq = oracle.ScalarQuantizer(sq.min_vals, sq.max_vals, sq.scale, 0.0)
            oi, od, oc = osq.batch_search(queries[:s], codes[:ns], k)
            extra["parity"] = {"queries_checked": s, "rows": ns, "ids_equal": bool(np.array_equal(gi, oi)),
                               "dists_bit_equal": bool(np.array_equal(bits(gd), bits(od)))}
            small.close()
        emit("sq8", f"SQ8 asymmetric scan over {n}x{dim} codes, k={k}", nq, sec, extra)
    ix.close()


def pq_case(ctx, dev, n, dim, m_sub, nq, k, steps, oracle=None, build_timing=True):
    rng = np.random.default_rng(17)
    books = rng.standard_normal((m_sub, 256, dim // m_sub), dtype=F)
    pq = na.quantization.ProductQuantizer(books)
    codes = rng.integers(0, 256, size=(n, m_sub), dtype=np.uint8)
    queries = rng.standard_normal((nq, dim), dtype=F)
    ix = na.quantization.PqIndex(ctx, pq, codes)
    dq = torch.from_numpy(queries).to(dev)
    for prec in ("bf16", "fp32"):
        ctx.set_profiling(True)
        ctx.reset_metrics()
        sec, out = timed(lambda: search_device(ix._h, dq, k, precision=prec), steps, 1)
        mm = ctx.metrics()
        ctx.set_profiling(False)
        scan_ms = mm["scan_ms_total"] / max(mm["scan_launches"], 1)
        if prec == "bf16":  # the decoded rows (bf16, dim values each) stream once per 256-query block
            byts = (-(-nq // 256)) * n * dim * 2
            extra = {"first_pass": "bf16 decoded rows on the matrix cores, table arithmetic for the candidates",
                     "scan_ms": round(scan_ms, 3), "bytes_scanned": byts, "scan_GBps": round(byts / scan_ms / 1e6, 1),
                     "code_bytes_per_row": m_sub, "decoded_row_bytes": 2 * dim,
                     "hbm_held_by_the_decoded_copy_GB": round(n * dim * 2 / 1e9, 2),
                     "frac_of_hbm_peak": round(byts / scan_ms / 1e6 / HBM_PEAK_GBS, 4),
                     "scan_tflops": round(2.0 * nq * n * dim / scan_ms / 1e9, 1),
                     "roofline": roofline("hbm", (-(-nq // 256)) * n * m_sub / scan_ms / 1e6, "GB/s", "pq: bf16 scan of the decoded rows",
                                          "m bytes per vector, once per query block (the route streams 2*D bytes of decoded row instead)", scan_ms),
                     "roofline_mfma": roofline("mfma", 2.0 * nq * n * dim / scan_ms / 1e9, "TFLOP/s", "flat_scan_bf16 over the decoded rows",
                                               "2*Q*N*D flop per batch", scan_ms),
                     "host_pointer_qps": host_api_qps(ix._h, queries, k, precision=prec),
                     "concurrent_callers_qps": ({str(t): concurrent_qps(ix._h, dq, k, t, max(steps, 4), precision=prec) for t in (2, 4)}
                                                if CONCURRENCY else None),
                     "fallback_per_batch": mm["fallback_queries"] / (steps + 1)}
        else:
            qg = 8  # queries whose interleaved tables a workgroup keeps resident (PQ_QB, csrc/nh_kernels.h)
            byts = (-(-nq // qg)) * n * m_sub  # every query group streams the code matrix once
            lookups = nq * n * m_sub / sec
            # a lookup is an LDS gather: 16-byte reads fetch the entries of 4 queries, 256 B per clock and CU when no two lanes
            # of a 16-lane phase share a four-bank slot (the kernel's lane skew makes the slot the subspace: no conflicts)
            lds_peak = 256 * 64 * 2.4e9
            extra = {"first_pass": "f32 distance-table scan", "queries_per_workgroup": qg, "bytes_scanned": byts,
                     "GBps_over_whole_search": round(byts / sec / 1e9, 1),
                     "frac_of_hbm_peak": round(byts / sec / 1e9 / HBM_PEAK_GBS, 4),
                     "roofline": roofline("hbm", byts / sec / 1e9, "GB/s", "pq_scan_kernel (distance tables in LDS)",
                                          "m bytes per vector, once per group of 8 queries", sec * 1e3),
                     "table_lookups_per_s": round(lookups / 1e12, 3),
                     "lds_gather": {"achieved_lookups_per_s": round(lookups / 1e12, 3), "conflict_free_peak": round(lds_peak / 1e12, 1),
                                    "unit": "1e12 lookups/s", "frac": round(lookups / lds_peak, 4),
                                    "note": "conflict-free gather (lane skew); beside it the kernel issues 10 vector instructions per 8 lookups and refills its 128 KB of tables once per block of rows"}}
        if oracle is not None:
            s, ns = 6, 500_000
            small = na.quantization.PqIndex(ctx, pq, codes[:ns])
            gi, gd, gc = small.search(queries[:s], k, precision=prec)
            opq = oracle.ProductQuantizer(books, np.full(m_sub, 256, dtype=np.uint32), dim, m_sub)
            oi, od, oc = opq.batch_search(queries[:s], codes[:ns], k)
            extra["parity"] = {"queries_checked": s, "rows": ns, "ids_equal": bool(np.array_equal(gi, oi)),
                               "dists_bit_equal": bool(np.array_equal(bits(gd), bits(od)))}
            small.close()
        emit("pq", f"PQ m={m_sub} asymmetric scan over {n}x{m_sub} codes (dim {dim}), k={k}", nq, sec, extra)
    ix.close()
    if not build_timing:
        return
    # codebook training and encoding on the device (Gaussian rows; the scans above use random codes and books)
    sample = rng.standard_normal((65536, dim), dtype=F)
    t0 = time.perf_counter()
    tpq = na.quantization.ProductQuantizer.train_device(ctx, sample, m_sub, seed=5)
    train_s = time.perf_counter() - t0
    rows = rng.standard_normal((min(n, 1_000_000), dim), dtype=F)
    t0 = time.perf_counter()
    tpq.encode_device(ctx, rows)
    line = {"path": "pq_build", "train_s_65536_rows": round(train_s, 2),
            "encode_rows_per_s_incl_pcie": round(rows.shape[0] / (time.perf_counter() - t0), 1)}
    if COLLECT is not None:
        COLLECT.append(line)
    else:
        print(json.dumps(line), flush=True)


def binary_case(ctx, dev, n, dim, nq, k, steps, oracle=None):
    rng = np.random.default_rng(19)
    bq = na.quantization.BinaryQuantizer(np.zeros(dim, dtype=F))
    nbytes = (dim + 7) // 8
    codes = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
    queries = rng.standard_normal((nq, dim), dtype=F)
    ix = na.quantization.BinaryIndex(ctx, bq, codes)
    dq = torch.from_numpy(queries).to(dev)
    ctx.set_profiling(True)
    ctx.reset_metrics()
    sec, out = timed(lambda: search_device(ix._h, dq, k), steps, 1)
    m = ctx.metrics()
    ctx.set_profiling(False)
    scan_ms = m["scan_ms_total"] / max(m["scan_launches"], 1)
    qt = 16 if dim + 1 <= 900 else 4
    scans = m["binary_code_scans"] / max(m["scan_launches"], 1)  # what the library did: histogram + collect, + a tie count when needed
    byts = scans * (-(-nq // qt)) * n * nbytes  # every scan streams the codes once per query tile
    extra = {"scan_ms": round(scan_ms, 3), "code_scans_per_batch": scans, "bytes_scanned": int(byts),
             "scan_GBps": round(byts / scan_ms / 1e6, 1), "frac_of_hbm_peak": round(byts / scan_ms / 1e6 / HBM_PEAK_GBS, 4),
             "xor_popcount_words_per_s": round(scans * nq * n * ((dim + 63) // 64) / scan_ms / 1e9, 2)}
    if oracle is not None:
        s, ns = 8, 500_000
        small = na.quantization.BinaryIndex(ctx, bq, codes[:ns])
        gi, gd, gc = small.search(queries[:s], k)
        oi, od, oc = oracle.BinaryQuantizer(bq.thresholds).batch_search(queries[:s], codes[:ns], k)
        extra["parity"] = {"queries_checked": s, "rows": ns, "ids_equal": bool(np.array_equal(gi, oi)),
                           "dists_bit_equal": bool(np.array_equal(bits(gd), bits(od)))}
        small.close()
    emit("binary", f"binary codes, Hamming scan over {n}x{dim} bits, k={k}", nq, sec, extra)
    ix.close()


def hnsw_case(ctx, dev, n, dim, nq, k, threads, oracle, data="gaussian", build_budget_s=150.0):
    """CPU baseline of the reference's default search path (src/collection/batch.rs:32-52 over src/indexing/hnsw.rs,
    m = 16, ef_construction = 200, ef_search = 50), restated in oracle/needle_hnsw.c.  The build is sequential as in the
    reference (insert takes &mut self) and runs until `n` rows are in or `build_budget_s` seconds have passed -- the
    searches (parallel over the queries, each paying the reference's per-neighbour distance calls) and the exact search
    they are compared with then cover the rows that are in."""
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    base = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
    queries = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
    if data == "clustered":  # a mixture of 1000 clusters: the structure real embeddings have and i.i.d. rows lack
        centres = torch.randn((1000, dim), generator=g, device=dev, dtype=torch.float32)
        base = centres[torch.randint(0, 1000, (n,), generator=g, device=dev)] + 0.5 * base
        queries = centres[torch.randint(0, 1000, (nq,), generator=g, device=dev)] + 0.5 * queries
    base = torch.nn.functional.normalize(base, dim=1).contiguous()
    queries = torch.nn.functional.normalize(queries, dim=1).contiguous()
    hb, hq = base.cpu().numpy(), queries.cpu().numpy()
    h = oracle.Hnsw(hb, oracle.COSINE)
    t0 = time.perf_counter()
    built = 0
    while built < n and time.perf_counter() - t0 < build_budget_s:
        step = min(2048, n - built)  # in pieces, with a progress line: a sequential build of minutes is silent otherwise
        h.insert_range(built, step)
        built += step
        print(f"hnsw build: {built}/{n} rows, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    build_s = time.perf_counter() - t0
    ix = na.FlatIndex(ctx, base[:built].contiguous(), na.DistanceFunction.Cosine)
    sec_gpu, out = timed(lambda: search_device(ix._h, queries, k, precision="bf16"), 3, 1)
    gt = out[0].cpu().numpy()
    for ef in (50, 200, 800):
        h.set_ef_search(ef)
        t0 = time.perf_counter()
        ids, _, _ = h.batch_search(hq, k, threads=threads)
        sec = time.perf_counter() - t0
        recall = float(np.mean([len(set(ids[i].tolist()) & set(gt[i].tolist())) / k for i in range(nq)]))
        emit("hnsw_cpu", f"reference HNSW (m=16, ef_construction=200, ef_search={ef}) restated on the CPU, {built}x{dim} cosine ({data} rows), k={k}",
             nq, sec, {"recall_at_k": round(recall, 4), "threads": threads, "rows": built, "rows_wanted": n,
                       "build_budget_s": build_budget_s, "build_budget_hit": built < n,
                       "build_s": round(build_s, 1), "build_ms_per_insert": round(build_s / max(built, 1) * 1e3, 3),
                       "build": "sequential; distances of the unvisited neighbours scored together (bit-identical graph)",
                       "library_exact_ms_same_rows": round(sec_gpu * 1e3, 3), "library_exact_qps_same_rows": round(nq / sec_gpu, 1),
                       "parity": "graph unpinned (the reference's levels are unseeded); properties pinned by its tests"})
    h.close()
    ix.close()


def run_paths(ctx, dev, want, steps=2, oracle=None, scale=1.0, hnsw_rows=100_000, cpu_threads=16, hnsw_budget_s=150.0):
    """The legs bench.py folds into its JSON line (`paths`): every entry carries its own roofline object computed
    from algorithmic bytes / flops, a parity sample against the oracle and the host-pointer (PCIe-inclusive) rate."""
    global COLLECT
    COLLECT = []
    sc = scale
    try:
        if "flat10m" in want:
            flat_case(ctx, dev, "flat10m", int(10_000_000 * sc), 384, 10_000, 10, na.DistanceFunction.Cosine, steps, oracle=oracle,
                      normalise=True, precisions=("bf16",), parity_queries=16)
        if "c1" in want:
            flat_case(ctx, dev, "c1", int(100_000 * sc), 128, 1000, 10, na.DistanceFunction.Euclidean, steps, oracle=oracle,
                      precisions=("bf16",))
        if "c5" in want:
            flat_case(ctx, dev, "c5", int(1_000_000 * sc), 768, 1000, 100, na.DistanceFunction.DotProduct, steps, filt_frac=0.1,
                      oracle=oracle, precisions=("bf16",), parity_queries=16)
        if "ivf" in want:
            ivf_case(ctx, dev, int(1_250_000 * sc), 384, max(16, int(4096 * sc)), 32, 10_000, 10, steps, oracle=oracle)
        if "ivf_clustered" in want:
            ivf_case(ctx, dev, int(1_250_000 * sc), 384, max(16, int(4096 * sc)), 32, 10_000, 10, steps, oracle=oracle, data="clustered")
        if "sq8" in want:
            sq8_case(ctx, dev, int(6_250_000 * sc), 384, 1000, 10, steps, oracle=oracle)
        if "pq" in want:
            pq_case(ctx, dev, int(6_250_000 * sc), 384, 48, 1000, 10, steps, oracle=oracle, build_timing=False)
        if "hnsw" in want and oracle is not None:
            hnsw_case(ctx, dev, hnsw_rows, 384, 1000, 10, cpu_threads, oracle, build_budget_s=hnsw_budget_s)
        return COLLECT
    finally:
        COLLECT = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", default="c1,c5,ivf,sq8,pq,binary")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the row counts (smoke runs)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--concurrency", action="store_true", help="also time 2 and 4 host threads searching one index at once")
    ap.add_argument("--hnsw-rows", type=int, default=100_000, help="rows of the hnsw path's sample (the build is sequential)")
    ap.add_argument("--hnsw-seconds", type=float, default=150.0, help="time budget of that build")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--hnsw-data", default="gaussian", choices=["gaussian", "clustered"])
    ap.add_argument("--ivf-centroids", default="trained", choices=["trained", "sampled"],
                    help="trained: Lloyd iterations on a sample (balanced lists); sampled: raw rows (hub lists)")
    args = ap.parse_args()
    global CONCURRENCY
    CONCURRENCY = args.concurrency
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = na.Context(0)
    ctx.apply_tuning_spec(os.environ.get("NEEDLE_BENCH_TUNE"))  # A/B sweeps: planning overrides, e.g. splits=64,pre_tiles=16
    oracle = None
    if not args.no_parity:
        from oracle import oracle as orc  # checker only

        orc.build()
        oracle = orc
    sc = args.scale
    want = args.paths.split(",")
    if "c1" in want:
        flat_case(ctx, dev, "c1", int(100_000 * sc), 128, 1000, 10, na.DistanceFunction.Euclidean, args.steps, oracle=oracle)
    if "c5" in want:
        flat_case(ctx, dev, "c5", int(1_000_000 * sc), 768, 1000, 100, na.DistanceFunction.DotProduct, args.steps,
                  filt_frac=0.1, oracle=oracle)
    if "ivf" in want:
        ivf_case(ctx, dev, int(1_250_000 * sc), 384, max(16, int(4096 * sc)), 32, 10_000, 10, args.steps, oracle=oracle,
                 centroids=args.ivf_centroids)
    if "ivf_clustered" in want:
        ivf_case(ctx, dev, int(1_250_000 * sc), 384, max(16, int(4096 * sc)), 32, 10_000, 10, args.steps, oracle=oracle,
                 centroids=args.ivf_centroids, data="clustered")
    if "sq8" in want:
        sq8_case(ctx, dev, int(6_250_000 * sc), 384, 1000, 10, args.steps, oracle=oracle)
    if "pq" in want:
        pq_case(ctx, dev, int(6_250_000 * sc), 384, 48, 1000, 10, args.steps, oracle=oracle)
    if "binary" in want:
        binary_case(ctx, dev, int(6_250_000 * sc), 384, 1000, 10, args.steps, oracle=oracle)
    if "hnsw" in want:
        if oracle is None:
            raise SystemExit("the hnsw path is the CPU restatement itself: run without --no-parity")
        hnsw_case(ctx, dev, args.hnsw_rows, 384, 1000, 10, args.cpu_threads, oracle, data=args.hnsw_data,
                  build_budget_s=args.hnsw_seconds)
    ctx.close()


if __name__ == "__main__":
    main()
