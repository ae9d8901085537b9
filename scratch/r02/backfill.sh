#!/bin/bash
# A/B of the reserved-CU backfill (SMN_BACKFILL): parity subset first, then the C4 step time per setting.
set -e
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "chol or allgather_part" 2>&1 | tail -3
for bf in 0 0.5 0.8 1.0; do for ch in 4.5 7; do
  [ "$bf" = 0 ] && [ "$ch" = 7 ] && continue
  echo "== SMN_BACKFILL=$bf SMN_BACKFILL_CHAIN=$ch"
  SMN_BACKFILL=$bf SMN_BACKFILL_CHAIN=$ch timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline'].get('cholesky_wall_ms'), j['result'])"
done; done
