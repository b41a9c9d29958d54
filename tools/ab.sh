#!/bin/bash
# A/B timing of library variants on the headline workload: tools/ab.sh name[:ENV=VAL+ENV=VAL...] ...
# (planning overrides go through the harness: main:NEEDLE_BENCH_TUNE=stages=15:30,pre_tiles=6)
# (name "main" = needle_amd/lib/libneedle_hip.so, otherwise needle_amd/lib/variants/libneedle_hip_<name>.so)
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr '+' ' ')
  lib=$PWD/needle_amd/lib/libneedle_hip.so
  [ "$name" != main ] && lib=$PWD/needle_amd/lib/variants/libneedle_hip_$name.so
  out=$(env $envs NEEDLE_HIP_LIBRARY=$lib timeout -k 10 120 python bench.py --paths none --cpu-seconds 0 --secondary-steps 0 --steps 10 $AB_ARGS 2>/dev/null); rc=$?
  # 3 = parity mismatch (expected from timing-only variants that skip work); the line is still printed
  [ $rc -ne 0 ] && [ $rc -ne 3 ] && { echo "$spec FAILED rc=$rc"; exit 1; }
  echo "$spec $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms/step", d["ms_per_step"], "scan_ms", d["roofline"]["launch_ms"], "frac", d["roofline"]["frac"])')"
done
