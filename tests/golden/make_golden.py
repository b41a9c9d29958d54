#!/usr/bin/env python3
"""Writes tests/golden/hotpath_golden.npz: expected outputs of the search path on the reference's own
deterministic inputs (gen_vector, src/recall_benchmark.rs:171-180; base seed 42, query seed
42 + 999_999 as src/recall_benchmark.rs:237-242), computed by the CPU oracle.

The reference is Rust and cannot run where this was authored, so the expected values come from the
oracle's restatement (itself pinned by the reference's known-answer tests); the fixture freezes them:
the CPU suite checks that the oracle still reproduces the file, the GPU suite checks the HIP path
against the file without the oracle in the loop.

    python tests/golden/make_golden.py        # regenerate (only when the oracle changes on purpose)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

PATH = os.path.join(HERE, "hotpath_golden.npz")
BASE_SEED, QUERY_SEED = 42, 42 + 999_999
# metric numbering of the C ABI (include/needle_hip.h) == the oracle's
METRICS = {"cosine": 0, "cosine_normalized": 1, "euclidean": 2, "dot": 3, "manhattan": 4, "hamming": 5, "chebyshev": 6}
FLAT = {"n": 3000, "dim": 128, "nq": 16, "k": 10}
WIDE = {"n": 2000, "dim": 384, "nq": 8, "k": 10}
IVF = {"n": 6000, "dim": 64, "nq": 24, "k": 10, "nlist": 16, "nprobe": 4}
ADC = {"n": 4000, "dim": 32, "nq": 12, "k": 10, "m": 4}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def inputs(spec):
    return (orc.gen_vectors(BASE_SEED, 0, spec["n"], spec["dim"]), orc.gen_vectors(QUERY_SEED, 0, spec["nq"], spec["dim"]))


def compute() -> dict:
    orc.build()
    out = {}
    base, queries = inputs(FLAT)
    masks = np.random.default_rng(5)
    valid = masks.random(FLAT["n"]) > 0.2
    filt = masks.random(FLAT["n"]) < 0.3
    out["flat_valid"], out["flat_filter"] = valid, filt
    for name, m in METRICS.items():
        b, q = (orc.normalize_rows(base), orc.normalize_rows(queries)) if name == "cosine_normalized" else (base, queries)
        ids, d, c = orc.flat_batch_search(q, b, FLAT["k"], m)
        out[f"flat_{name}_ids"], out[f"flat_{name}_dist_bits"], out[f"flat_{name}_counts"] = ids, bits(d), c
        ids, d, c = orc.flat_batch_search(q, b, FLAT["k"], m, valid=valid, filt=filt)
        out[f"flat_{name}_masked_ids"], out[f"flat_{name}_masked_dist_bits"] = ids, bits(d)
    base, queries = inputs(WIDE)
    for name in ("cosine", "euclidean"):
        ids, d, _ = orc.flat_batch_search(queries, base, WIDE["k"], METRICS[name])
        out[f"wide_{name}_ids"], out[f"wide_{name}_dist_bits"] = ids, bits(d)
    # IVF-Flat: centroids from the oracle's seeded k-means (the reference trains from thread_rng), kept in the file
    base, queries = inputs(IVF)
    cents = orc.ivf_kmeans(base[:1600], IVF["nlist"], max_iter=10, seed=3)
    assign = orc.ivf_assign(base, cents)
    offsets, order = orc.ivf_lists_from_assignment(assign, cents.shape[0])
    ids, d, c = orc.ivf_batch_search(queries, base, cents, offsets, order, IVF["nprobe"], IVF["k"])
    out["ivf_centroids"], out["ivf_assign"] = cents, assign
    out["ivf_probe"] = np.stack([orc.ivf_probe(q, cents, IVF["nprobe"]) for q in queries])
    out["ivf_ids"], out["ivf_dist_bits"], out["ivf_counts"] = ids, bits(d), c
    # SQ8 and PQ asymmetric scans
    base, queries = inputs(ADC)
    sq = orc.ScalarQuantizer.train(base[:1000])
    codes = sq.quantize(base)
    ids, d, _ = sq.batch_search(queries, codes, ADC["k"])
    out["sq_min"], out["sq_max"], out["sq_scale"] = sq.min_vals, sq.max_vals, sq.scale
    out["sq_codes_head"] = codes[:64]
    out["sq_ids"], out["sq_dist_bits"] = ids, bits(d)
    pq = orc.ProductQuantizer.train(base[:1000], ADC["m"], seed=3)
    pcodes = pq.encode(base)
    ids, d, _ = pq.batch_search(queries, pcodes, ADC["k"])
    out["pq_codebooks"], out["pq_codes"] = pq.codebooks, pcodes
    out["pq_ids"], out["pq_dist_bits"] = ids, bits(d)
    return out


if __name__ == "__main__":
    np.savez_compressed(PATH, **compute())
    print(PATH, os.path.getsize(PATH), "bytes")
