#!/usr/bin/env python3
"""Same-box sweep of planning overrides over the PQ distance-table scan (precision fp32):
    python tools/pq_sweep.py "" splits=2 splits=8 splits=32,debug=1 ...
Each argument is a tuning spec (key=value,...; "" = the built-in plan); the index is built once."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import needle_amd as na  # noqa: E402
from tools.bench_paths import search_device, timed  # noqa: E402

F = np.float32


def main():
    specs = sys.argv[1:] or [""]
    n, dim, m_sub, nq, k = int(os.environ.get("PQ_ROWS", 6_250_000)), 384, 48, int(os.environ.get("PQ_QUERIES", 1000)), 10
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ctx = na.Context(0)
    rng = np.random.default_rng(17)
    books = rng.standard_normal((m_sub, 256, dim // m_sub), dtype=F)
    pq = na.quantization.ProductQuantizer(books)
    codes = rng.integers(0, 256, size=(n, m_sub), dtype=np.uint8)
    queries = rng.standard_normal((nq, dim), dtype=F)
    ix = na.quantization.PqIndex(ctx, pq, codes)
    dq = torch.from_numpy(queries).to(dev)
    for spec in specs:
        ctx.set_tuning(reset=0)
        ctx.apply_tuning_spec(spec or None)
        ctx.set_profiling(True)
        ctx.reset_metrics()
        sec, _ = timed(lambda: search_device(ix._h, dq, k, precision="fp32"), 3, 1)
        mm = ctx.metrics()
        ctx.set_profiling(False)
        scan_ms = mm["scan_ms_total"] / max(mm["scan_launches"], 1)
        print(f"{spec or 'default':32s} ms/batch {sec * 1e3:8.3f}  tables+scan {scan_ms:8.3f}  lookups/s {nq * n * m_sub / sec / 1e12:6.2f}e12", flush=True)
    ix.close()


if __name__ == "__main__":
    t0 = time.time()
    main()
