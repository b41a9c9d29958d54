#!/bin/bash
for m in 0,4 0,32 0,2 0,8 0,1 1,6; do
  out=$(NEEDLE_HIP_DEBUG_IVF_STAGES=$m timeout -k 10 200 python tools/bench_paths.py --paths ivf --steps 5 --no-parity 2>/dev/null | tail -1) || exit 1
  echo "stages=$m ivf $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d.get("ms_per_batch"), d.get("qps"), d.get("list_scan_ms"))')"
done
