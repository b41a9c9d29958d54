#!/bin/bash
# Same-box sweep of planning overrides on one shard size of the strong-scaling curve: ROWS=125000 tools/ab_rows_tune.sh spec ...
for t in "$@"; do
NEEDLE_BENCH_TUNE=$t timeout -k 10 120 python bench.py --rows ${ROWS:-125000} --paths none --cpu-seconds 0 --secondary-steps 0 --steps 20 --warmup 3 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d[\"config\"][\"rows_total\"], \"ms/step\", d[\"ms_per_step\"], \"scan_ms\", d[\"roofline\"][\"launch_ms\"], \"frac\", d[\"roofline\"][\"frac\"], 'parity', d['parity'])" || exit 1
done
exit 0
