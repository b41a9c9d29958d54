#!/usr/bin/env python3
"""The headline workload through the HOST-pointer entry point (needle_hip_batch_search): queries uploaded and results
downloaded over PCIe inside the timed region -- the rate a caller holding host buffers sees.  bench.py's value is the
device-resident rate; this one is reported beside it, never as the headline."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import needle_amd as na  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, dim, nq, k = 1_000_000, 384, 10_000, 10
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    base = torch.nn.functional.normalize(torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32), dim=1)
    queries = torch.nn.functional.normalize(torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32), dim=1)
    ctx = na.Context(0)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    hq = queries.cpu().numpy()
    for prec in ("bf16",):
        ix.search(hq, k, precision=prec)
        t0 = time.perf_counter()
        steps = 10
        for _ in range(steps):
            ids, dists, counts = ix.search(hq, k, precision=prec)
        sec = (time.perf_counter() - t0) / steps
        # the device-resident call on the same box, for the difference
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oc = torch.empty((nq,), dtype=torch.int32, device=dev)
        ix.search_device(queries, k, oi, od, oc, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ix.search_device(queries, k, oi, od, oc, precision=prec)
        torch.cuda.synchronize()
        sec_dev = (time.perf_counter() - t0) / steps
        same = bool(np.array_equal(ids.view(np.int64), oi.cpu().numpy()))
        print(json.dumps({"workload": f"exact flat {n}x{dim} cosine, {nq} queries, k={k}, first pass {prec}",
                          "host_api_ms": round(sec * 1e3, 3), "host_api_qps": round(nq / sec, 1),
                          "device_api_ms": round(sec_dev * 1e3, 3), "device_api_qps": round(nq / sec_dev, 1),
                          "host_buffers": "pageable numpy arrays (15.4 MB of queries in, 1.2 MB of results out)",
                          "same_ids": same}))
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
