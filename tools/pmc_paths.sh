#!/bin/bash
# One PMC pass over a path of tools/bench_paths.py, summed per kernel.  Usage: tools/pmc_paths.sh <path> <counter> [counter ...]
# (at most 8 SQ_, 4 TCC_, 2 GRBM_ counters per pass)
set -u
P=${1:-ivf}; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmc_path_$P
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc" -- \
    python3 tools/bench_paths.py --paths "$P" --no-parity --steps 1 > "$OUT/pmc.log" 2>&1 || { echo "pmc run failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(float)); calls=defaultdict(set)
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'] or 0); calls[k].add(r['Dispatch_Id'])
for k,cs in sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))[:14]:
    n=len(calls[k])
    print(f"{k}  dispatches={n}")
    print("    "+"  ".join(f"{c}={v/n:.4g}" for c,v in sorted(cs.items())))
PY
