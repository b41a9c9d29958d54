#!/bin/bash
# first-pass time of every search from process start (is the first searches' slowness page mapping of fresh workspace?)
for cfg in "X=0" "NEEDLE_HIP_DEBUG_POISON=1"; do
  echo "== $cfg"
  env $cfg NEEDLE_HIP_DEBUG=8 timeout -k 10 180 python bench.py --paths none --cpu-seconds 0 --secondary-steps 0 --steps 12 --warmup 0 2>&1 >/dev/null | grep "first pass bf16" | awk '{printf "%s ", $(NF-1)} END{print ""}'
done
