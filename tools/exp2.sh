#!/bin/bash
# Threshold-seeding experiments on the bf16 scan: sample size and seed rank.
set -u
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --precision bf16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/exp2.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d.get('parity'), d.get('fallback_queries_per_step'))"; grep "needle_hip" gpurun_out/exp2.err | tail -1; }
run NEEDLE_HIP_DEBUG=8 || exit 1
run NEEDLE_HIP_DEBUG=8 NEEDLE_HIP_DEBUG_KSEED=16 || exit 1
run NEEDLE_HIP_DEBUG=8 NEEDLE_HIP_DEBUG_PRE=30 NEEDLE_HIP_DEBUG_KSEED=16 || exit 1
run NEEDLE_HIP_DEBUG=8 NEEDLE_HIP_DEBUG_PRE=60 NEEDLE_HIP_DEBUG_KSEED=10 || exit 1
