#!/bin/bash
# Where does the bf16 scan spend its time?  Development switches, results are wrong by design.
set -u
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --precision bf16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/exp4.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
run NEEDLE_HIP_DEBUG=0 || exit 1
run NEEDLE_HIP_DEBUG=1 || exit 1
run NEEDLE_HIP_DEBUG=17 || exit 1
run NEEDLE_HIP_DEBUG=33 || exit 1
run NEEDLE_HIP_DEBUG=49 || exit 1
