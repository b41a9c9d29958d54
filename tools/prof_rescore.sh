#!/bin/bash
export TMPDIR=/tmp
for cfg in "X=0" "NEEDLE_HIP_DEBUG_NOSETONLY=1" "NEEDLE_HIP_DEBUG_RSCAP=8"; do
  rm -rf /tmp/prof_x; mkdir -p /tmp/prof_x
  env $cfg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python3 tools/bench_paths.py --paths ivf --no-parity --steps 4 > /tmp/prof_x/log 2>&1 || { echo "$cfg failed"; tail -3 /tmp/prof_x/log; exit 1; }
  echo "== $cfg"
  python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/prof_x/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r['Name'] for s in ('rescore','prune','dense_select','merge_select','seed_tau')):
            print(f"  {r['Name'][:70]:70s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.3f}")
PY
done
