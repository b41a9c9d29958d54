#!/bin/bash
# exact re-score: rows staged at a time / dimensions per block, on the IVF and filtered 768-d paths
run() {
  env "$@" AB_STEPS=20 bash tools/ab_paths.sh ivf,c5 main | grep bf16 | awk -v tag="$*" '{printf "%s | %s %s  ", tag, $2, $NF} END{print ""}'
}
run X=0
run NEEDLE_HIP_DEBUG_RSCAP=16
run NEEDLE_HIP_DEBUG_RSCAP=32 NEEDLE_HIP_DEBUG_RSDC=64
run NEEDLE_HIP_DEBUG_RSCAP=64 NEEDLE_HIP_DEBUG_RSDC=128
run NEEDLE_HIP_DEBUG_RSCAP=128 NEEDLE_HIP_DEBUG_RSDC=64
run NEEDLE_HIP_DEBUG_RSCAP=32 NEEDLE_HIP_DEBUG_RSDC=128
run NEEDLE_HIP_DEBUG_RSCAP=256 NEEDLE_HIP_DEBUG_RSDC=48
run X=0
