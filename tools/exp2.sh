#!/bin/bash
# Staged launches with threshold refresh vs one launch (NEEDLE_HIP_DEBUG bit 4 = single launch).
set -u
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --precision bf16 --steps 3 --warmup 1 --cpu-seconds 0 --secondary-steps 0 2>gpurun_out/exp2.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d.get('parity'), d.get('fallback_queries_per_step'))"; grep "needle_hip" gpurun_out/exp2.err | tail -1; }
run NEEDLE_HIP_DEBUG=8 || exit 1
run NEEDLE_HIP_DEBUG=12 || exit 1
run NEEDLE_HIP_DEBUG=1 || exit 1
run NEEDLE_HIP_DEBUG=5 || exit 1
