#!/bin/bash
# C5 path (1M x 768 dot, 10 % filter, k = 100, bf16 first pass) under planning overrides:
# tools/ab_c5_tune.sh "spec" ...   (spec as NEEDLE_BENCH_TUNE takes it; "-" = none)
for spec in "$@"; do
  t=$spec; [ "$spec" = "-" ] && t=""
  out=$(NEEDLE_BENCH_TUNE=$t timeout -k 10 200 python tools/bench_paths.py --paths c5 --steps 8 --no-parity 2>/dev/null) || { echo "$spec FAILED"; exit 1; }
  echo "$spec $(echo "$out" | python -c '
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if d.get("path")=="c5" and "bf16" in json.dumps(d.get("first_pass", d.get("precision",""))):
        print("ms/batch", d["ms_per_batch"], "scan_ms", d.get("scan_ms")); break
')"
done
