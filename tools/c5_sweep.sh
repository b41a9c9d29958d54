#!/bin/bash
# C5 (1M x 768 dot product, 10 % filter, k = 100, 1K queries): first-pass schedule variants, 20 steps each
run() {
  out=$(env "$@" timeout -k 10 200 python tools/bench_paths.py --paths c5 --steps 20 --no-parity 2>/dev/null | grep bf16 | tail -1) || { echo "$* FAILED"; exit 1; }
  echo "$* $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/batch", d.get("ms_per_batch"), "scan_ms", d["roofline"]["launch_ms"], "frac", d["roofline"]["frac"])')"
}
run X=0
run NEEDLE_HIP_DEBUG_PRE=16
run NEEDLE_HIP_DEBUG_PRE=24
run NEEDLE_HIP_DEBUG_STAGES=5,15,30
run NEEDLE_HIP_DEBUG_STAGES=5,15,30 NEEDLE_HIP_DEBUG_PRE=16
run NEEDLE_HIP_DEBUG_SPLITS=64
run NEEDLE_HIP_DEBUG_SPLITS=64 NEEDLE_HIP_DEBUG_PRE=16
run X=0
NEEDLE_HIP_DEBUG=8 timeout -k 10 200 python tools/bench_paths.py --paths c5 --steps 6 --no-parity 2>&1 >/dev/null | grep -i "first pass" | head -12
