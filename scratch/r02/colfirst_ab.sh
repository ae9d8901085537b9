#!/bin/bash
# Column-first updates of the chain (rest of every near / F0 / far update on a side stream): parity, then A/B.
set -e
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -3
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], j['result']['logpdf']))"
}
for round in 1 2 3; do
  for cf in 0 1; do echo "round $round C4 SMN_COL_FIRST=$cf"; SMN_COL_FIRST=$cf one --steps 20 --warmup 3; done
done
for mt in 32 64 96; do echo "C4 SMN_COL_FIRST_MAX_TILES=$mt"; SMN_COL_FIRST_MAX_TILES=$mt one --steps 20 --warmup 3; done
for cf in 0 1; do echo "C2 SMN_COL_FIRST=$cf"; SMN_COL_FIRST=$cf one --config c2 --steps 50 --warmup 5; done
for cf in 0 1; do echo "C5 SMN_COL_FIRST=$cf"; SMN_COL_FIRST=$cf one --config c5 --steps 4 --warmup 1; done
for cf in 0 1; do echo "f64 n8192 SMN_COL_FIRST=$cf"; SMN_COL_FIRST=$cf one --dtype f64 --n 8192 --steps 10 --warmup 2; done
