#!/bin/bash
# F0 cut into head (next outer panel's two tile columns, caller's stream) and rest (bulk stream in front of F1): parity, A/B.
set -e
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -3
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['result']['logpdf']))"
}
for round in 1 2 3; do
  for v in 0 1; do echo "round $round C4 SMN_F0_REST=$v"; SMN_F0_REST=$v one --steps 20 --warmup 3; done
done
for v in 0 1; do echo "C5 SMN_F0_REST=$v"; SMN_F0_REST=$v one --config c5 --steps 4 --warmup 1; done
for v in 0 1; do echo "f64 n8192 SMN_F0_REST=$v"; SMN_F0_REST=$v one --dtype f64 --n 8192 --steps 10 --warmup 2; done
for v in 0 1; do echo "N=12288 SMN_F0_REST=$v"; SMN_F0_REST=$v one --n 12288 --steps 10 --warmup 2; done
