#!/bin/bash
# A/B of library variants on the other paths: tools/ab_paths.sh <paths> name ...   ("main" = the built library)
paths=$1; shift
for name in "$@"; do
  lib=$PWD/needle_amd/lib/libneedle_hip.so
  [ "$name" != main ] && lib=$PWD/needle_amd/lib/variants/libneedle_hip_$name.so
  NEEDLE_HIP_LIBRARY=$lib timeout -k 10 300 python tools/bench_paths.py --paths $paths --steps ${AB_STEPS:-20} --no-parity 2>/dev/null > /tmp/ab_paths.out || { echo "$name FAILED"; exit 1; }
  python - "$name" <<'P'
import json, sys
for ln in open('/tmp/ab_paths.out'):
    ln = ln.strip()
    if not ln.startswith('{'):
        continue
    d = json.loads(ln)
    print(sys.argv[1], d.get('path'), str(d.get('first_pass', d.get('list_scan', '')))[:24], 'ms/batch', d.get('ms_per_batch'))
P
done
