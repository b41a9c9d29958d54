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
grep '"path"' "$OUT/trace.log" | cut -c1-300
