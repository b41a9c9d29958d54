#!/bin/bash
# exact re-score: rows staged at a time / dimensions per block, on the headline, IVF and filtered 768-d paths
run() {
  a=$(env "$@" AB_STEPS=20 bash tools/ab_paths.sh ivf,c5 main | grep bf16 | awk '{printf "%s %s  ", $2, $NF}')
  b=$(env "$@" bash tools/ab.sh main | awk '{print "headline", $3}')
  echo "$* | $a $b"
}
run X=0
run NEEDLE_HIP_DEBUG_RSCAP=32 NEEDLE_HIP_DEBUG_RSDC=128
run NEEDLE_HIP_DEBUG_RSCAP=32 NEEDLE_HIP_DEBUG_RSDC=64
run NEEDLE_HIP_DEBUG_RSCAP=48 NEEDLE_HIP_DEBUG_RSDC=96
run NEEDLE_HIP_DEBUG_RSCAP=24 NEEDLE_HIP_DEBUG_RSDC=128
run X=0
