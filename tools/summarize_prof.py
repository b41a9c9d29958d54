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
