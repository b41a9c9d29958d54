#!/bin/bash
# development experiment: scan kernel time under different switches
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['launch_ms'], d['roofline']['achieved'], d['fallback_queries_per_step'])"; }
for v in "$@"; do run $v; done
