#!/bin/bash
# filtered 768-d search: width of the threshold band against time and refinement passes
for w in 2 1.5 1 2; do
  out=$(NEEDLE_HIP_DEBUG_BAND=$w timeout -k 10 200 python tools/bench_paths.py --paths c5 --steps 20 --no-parity 2>/dev/null | grep bf16 | tail -1) || { echo "band $w FAILED"; exit 1; }
  ref=$(NEEDLE_HIP_DEBUG_BAND=$w NEEDLE_HIP_DEBUG=8 timeout -k 10 200 python tools/bench_paths.py --paths c5 --steps 4 --no-parity 2>&1 >/dev/null | grep -c "first pass fp32")
  echo "band=$w $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/batch", d.get("ms_per_batch"), "scan_ms", d["roofline"]["launch_ms"])') fp32_passes_in_6_batches_incl_fp32_leg=$ref"
done
