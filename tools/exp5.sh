#!/bin/bash
# A/B of bf16 scan builds in one job (same box): main library and needle_amd/lib/variants/*.so, two rounds each.
set -u
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --precision bf16 --steps 8 --warmup 2 --cpu-seconds 0 --secondary-steps 0 2>gpurun_out/exp5.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['achieved'])"; }
for round in 1 2; do
  run A=main || exit 1
  for v in needle_amd/lib/variants/*.so; do
    run NEEDLE_HIP_LIBRARY=$PWD/$v || exit 1
  done
done
