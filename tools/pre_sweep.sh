#!/bin/bash
# headline search: tiles per slice in the threshold sample, and the share of a slice in the first launch
for cfg in "X=0" "NEEDLE_HIP_DEBUG_PRE=4" "NEEDLE_HIP_DEBUG_PRE=6" "NEEDLE_HIP_DEBUG_PRE=12" "NEEDLE_HIP_DEBUG_PRE=4+NEEDLE_HIP_DEBUG_STAGES=5,25" "NEEDLE_HIP_DEBUG_PRE=6+NEEDLE_HIP_DEBUG_STAGES=7,25" "NEEDLE_HIP_DEBUG_STAGES=7,25" "X=0"; do
  bash tools/ab.sh main:$cfg
done
