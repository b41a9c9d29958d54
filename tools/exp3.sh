#!/bin/bash
# Instruction-cache behaviour of the bf16 scan with and without candidate pushes.
set -u
export TMPDIR=/tmp
for mode in 0 1; do
  OUT=gpurun_out/icache_$mode
  mkdir -p $OUT
  NEEDLE_HIP_DEBUG=$mode timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 bench.py --precision bf16 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/log.txt 2>&1 || { echo failed; tail -5 $OUT/log.txt; exit 1; }
  python3 - $OUT <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'flat_scan_bf16_kernel<0' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
print(sys.argv[1],dict(agg))
PY
done
