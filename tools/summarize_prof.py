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


# HBM bytes of the first-pass scan kernels per SEARCH (a search may stage its scan over several launches: the listing
# launches of one search are summed; the seeding pre-pass runs once per search and counts the searches).  FETCH_SIZE and
# WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) reads at 64 bytes,
# so it is doubled.  The clock the chip held under the kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration.
import hashlib
import json

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = {"flat_scan_bf16": ["needle_amd/csrc/flat_scan_bf16.hip", "needle_amd/csrc/nh_scan.h", "needle_amd/csrc/nh_select.h"],
           "flat_scan_f32": ["needle_amd/csrc/flat_scan.hip", "needle_amd/csrc/nh_scan.h", "needle_amd/csrc/nh_select.h"]}


def source_hash(tag):
    h = hashlib.sha256()
    for rel in SOURCES[tag]:
        with open(os.path.join(REPO, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def dispatches(sub, counter, kern):
    """{dispatch id: (value summed over its rows, duration ns)} of kernels whose name contains `kern`."""
    out = {}
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kern in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                    v, _ = out.get(r["Dispatch_Id"], (0.0, 0))
                    out[r["Dispatch_Id"]] = (v + float(r.get("Counter_Value", 0) or 0),
                                             int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0))
    return out


traffic = {}
for stem, tag in (("flat_scan_bf16x_kernel<", "flat_scan_bf16"), ("flat_scan_f32_kernel<", "flat_scan_f32")):
    fetch = dispatches("pmc_fetch", "FETCH_SIZE", stem + "0")
    write = dispatches("pmc_write", "WRITE_SIZE", stem + "0")
    gui = dispatches("pmc_fetch", "GRBM_GUI_ACTIVE", stem + "0")
    searches_f = len(dispatches("pmc_fetch", "FETCH_SIZE", stem + "1")) or 1
    searches_w = len(dispatches("pmc_write", "WRITE_SIZE", stem + "1")) or 1
    if not fetch or not write:
        continue
    fetch_kb = sum(v for v, _ in fetch.values()) / searches_f
    write_kb = sum(v for v, _ in write.values()) / searches_w
    cyc = sum(v for v, _ in gui.values())
    ns = sum(d for _, d in gui.values())
    traffic[tag] = {"hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
                    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                    "launches_per_search": len(fetch) / searches_f, "searches": [searches_f, searches_w],
                    "held_clock_ghz": round(cyc / 8.0 / ns, 4) if ns > 0 else None,
                    "kernel_ms_per_search_under_the_counters": round(ns / searches_f / 1e6, 3),
                    "source_sha256": source_hash(tag)}
print("== first-pass HBM traffic per search (FETCH_SIZE x 2 + WRITE_SIZE, bytes; staged launches of one search summed)")
print(json.dumps(traffic, indent=1))

if "--write-traffic" in sys.argv:
    # record for bench.py's roofline.traffic: only used there while the kernel sources still hash to source_sha256
    wl = {"rows": 1000000, "dim": 384, "queries": 10000, "k": 10, "metric": "cosine"}
    for a in sys.argv:
        if a.startswith("--workload="):
            wl.update(json.loads(a.split("=", 1)[1]))
    dst = os.path.join(REPO, "needle_amd", "measured_traffic.json")
    try:
        with open(dst) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        rec = {}
    for tag, t in traffic.items():
        rec[tag] = {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "launches_per_search": t["launches_per_search"],
                    "held_clock_ghz": t["held_clock_ghz"], "source_sha256": t["source_sha256"], "workload": wl,
                    "source": f"{os.path.basename(os.path.normpath(root))}: rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / WRITE_SIZE passes "
                              "(FETCH_SIZE doubled; the staged launches of one search summed)"}
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")
