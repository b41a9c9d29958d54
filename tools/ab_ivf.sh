#!/bin/bash
# A/B timing of library variants on the IVF path: tools/ab_ivf.sh name ...   ("main" = the built library)
for name in "$@"; do
  lib=$PWD/needle_amd/lib/libneedle_hip.so
  [ "$name" != main ] && lib=$PWD/needle_amd/lib/variants/libneedle_hip_$name.so
  out=$(NEEDLE_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_paths.py --paths ivf --steps 5 2>/dev/null | tail -1) || { echo "$name FAILED"; exit 1; }
  echo "$name $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/batch", d["ms_per_batch"], "list_scan_ms", d.get("list_scan_ms"), "parity", d["parity"]["ids_equal"], d["parity"]["dists_bit_equal"])')"
done
