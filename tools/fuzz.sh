#!/bin/bash
# the seeded random parity sweeps over fresh cases: tools/fuzz.sh <offset> ...
for off in "$@"; do
  NEEDLE_TEST_SEED_OFFSET=$off timeout -k 10 500 python -u -m pytest tests/test_random_sweep_gpu.py tests/test_random_sweep_indexes_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -4 | sed "s/^/[offset $off] /"
  [ ${PIPESTATUS[0]} -ne 0 ] && { echo "offset $off FAILED"; exit 1; }
done
exit 0
