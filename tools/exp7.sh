#!/bin/bash
set -u
for st in "0,4" "0,0" "1,4" "0,8"; do
  echo "== stages $st"; NEEDLE_HIP_DEBUG_IVF_STAGES=$st timeout -k 10 200 python3 tools/bench_paths.py --paths ivf --no-parity --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16', d['ms_per_batch'], 'fp32', d['fp32_list_scan_ms_per_batch'])"
done
