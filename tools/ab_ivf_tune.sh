#!/bin/bash
# IVF path under planning overrides: tools/ab_ivf_tune.sh "spec" ...   (spec as NEEDLE_BENCH_TUNE takes it; "-" = none)
for spec in "$@"; do
  t=$spec; [ "$spec" = "-" ] && t=""
  out=$(NEEDLE_BENCH_TUNE=$t timeout -k 10 200 python tools/bench_paths.py --paths ivf --steps 8 --no-parity 2>/dev/null | tail -1) || { echo "$spec FAILED"; exit 1; }
  echo "$spec $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/batch", d["ms_per_batch"], "list_scan_ms", d.get("list_scan_ms"))')"
done
