#!/bin/bash
# Profile the default bench workload: kernel trace + stats, then one PMC pass per counter group.
# Usage (on the GPU box, from the repo root): tools/prof_flat.sh <tag> [extra bench args]
set -u
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --paths none "$@" > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --paths none "$@" > "$OUT/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --paths none "$@" > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- \
    python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --paths none "$@" > "$OUT/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
python3 tools/summarize_prof.py "$OUT" ${WRITE_TRAFFIC:+--write-traffic} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
