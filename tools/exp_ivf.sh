#!/bin/bash
# IVF seed-sample sweep (development)
set -u
for rows in 128 64 256 32; do
  echo "== seed rows $rows"
  NEEDLE_HIP_DEBUG_IVF_SEED_ROWS=$rows NEEDLE_HIP_DEBUG=8 timeout -k 10 200 python3 tools/bench_paths.py --paths ivf --steps 3 --no-parity 2> gpurun_out/exp_ivf.err | cut -c1-200 || exit 1
  grep ivf: gpurun_out/exp_ivf.err | tail -1
done
