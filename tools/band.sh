#!/bin/bash
# width of the threshold band (score uncertainties below the k-th score): time and undecided queries of the headline search
for w in 2 1.5 1 0.5 0 2; do
  out=$(NEEDLE_HIP_DEBUG_BAND=$w timeout -k 10 120 python bench.py --paths none --cpu-seconds 0 --secondary-steps 0 --steps 10 2>/dev/null | tail -1)
  echo "band=$w $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "scan_ms", d["roofline"]["launch_ms"], "refined/step", d["refined_queries_per_step"], "fallback/step", d["fallback_queries_per_step"])')"
done
