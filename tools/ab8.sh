#!/bin/bash
# first-pass time per search from the library's own debug line (robust against refinement passes of timing-only variants)
for name in "$@"; do
  lib=$PWD/needle_amd/lib/libneedle_hip.so
  [ "$name" != main ] && lib=$PWD/needle_amd/lib/variants/libneedle_hip_$name.so
  NEEDLE_BENCH_TUNE=debug=8 NEEDLE_HIP_LIBRARY=$lib timeout -k 10 180 python bench.py --paths none --cpu-seconds 0 --secondary-steps 0 --steps 8 --warmup 2 2>&1 >/dev/null | grep "first pass bf16" | awk -v n=$name '{ms[NR]=$(NF-1); l=$7} END{s=0;c=0;for(i=4;i<=NR;i++){s+=ms[i];c++} printf "%s first-pass bf16 ms: mean(last %d)=%.3f  listed/query=%s\n", n, c, s/c, l}'
done
