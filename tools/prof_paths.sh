#!/bin/bash
# Kernel trace of one path of tools/bench_paths.py.  Usage: tools/prof_paths.sh <path> [extra args]
set -u
P=${1:-ivf}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_path_$P
mkdir -p "$OUT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 tools/bench_paths.py --paths "$P" --no-parity --steps 2 "$@" > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_stats.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']} total_ms={float(r['TotalDurationNs'])/1e6:.3f} avg_ms={float(r['AverageNs'])/1e6:.3f} pct={r['Percentage']}")
PY
python3 - "$OUT" <<'PY'
# kernel timeline of the LAST batch (everything after the last gap of more than 2 ms between kernels)
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_trace.csv',recursive=True):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
    start=0
    for i in range(1,len(rows)):
        if int(rows[i]['Start_Timestamp'])-int(rows[i-1]['End_Timestamp'])>2_000_000: start=i
    last=rows[start:]
    t0=int(last[0]['Start_Timestamp'])
    print(f"== last batch: {len(last)} kernels, {(int(last[-1]['End_Timestamp'])-t0)/1e6:.3f} ms first start to last end")
    for r in last:
        print(f"  +{(int(r['Start_Timestamp'])-t0)/1e6:7.3f} ms  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f} ms  {r['Kernel_Name'][:100]}")
PY
grep '"path"' "$OUT/trace.log" | cut -c1-1500
