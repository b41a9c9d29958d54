#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


for f in find("trace", "*kernel_stats.csv"):
    print(f"== kernel stats ({os.path.relpath(f, root)})")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print(f"  {r.get('Name','')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
              f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        print(f"== counters ({os.path.relpath(f, root)})")
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(int)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")[:60]
                agg[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0) or 0)
        for k, cs in agg.items():
            if "flat_scan" not in k and "rescore" not in k and "merge" not in k:
                continue
            print("  " + k)
            for c, v in sorted(cs.items()):
                print(f"      {c:28s} {v:.6g}")


# HBM bytes per launch of the first-pass scan kernels: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
# FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) reads at 64 bytes, so it is doubled.
import json

traffic = {}
for kern, tag in (("flat_scan_bf16x_kernel<0", "flat_scan_bf16"), ("flat_scan_bf16_kernel<0", "flat_scan_bf16"),
                  ("flat_scan_f32_kernel<0", "flat_scan_f32")):
    if tag in traffic:
        continue
    fetch = write = 0.0
    n_f = n_w = 0
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in find(sub, "*counter_collection.csv"):
            per_dispatch = defaultdict(float)
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if kern in r.get("Kernel_Name", "") and r.get("Counter_Name") == name:
                        per_dispatch[r.get("Dispatch_Id")] += float(r.get("Counter_Value", 0) or 0)
            if per_dispatch:
                avg = sum(per_dispatch.values()) / len(per_dispatch)
                if name == "FETCH_SIZE":
                    fetch, n_f = avg, len(per_dispatch)
                else:
                    write, n_w = avg, len(per_dispatch)
    if n_f and n_w:
        traffic[tag] = {"hbm_bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
                        "fetch_size_kb": fetch, "write_size_kb": write, "launches": [n_f, n_w]}
print("== first-pass HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, bytes)")
print(json.dumps(traffic, indent=1))
