#!/usr/bin/env python3
"""Timeline of the kernels around the last launch matching a name in a rocprofv3 kernel trace.
Usage: trace_tail.py <dir with trace/**/_kernel_trace.csv> [count] [name substring]"""
import csv, glob, os, sys
root = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
name = sys.argv[3] if len(sys.argv) > 3 else "rescore_finalize"
fs = sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
last = idx[-1]
t_end = int(rows[last]["End_Timestamp"])
for r in rows[max(0, last - count):last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f %8.3f ms grid=%6d %s" % ((s - t_end) / 1e6, (e - s) / 1e6,
                                        int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Kernel_Name"][:70]))
