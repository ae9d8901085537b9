#!/bin/bash
# panelh_kernel (block updates on helper waves): bounded first run, the parity suite, then A/B timings.
set -e
timeout -k 10 150 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky_shift_and_not_pd or test_lml_gaussian" 2>&1 | tail -3
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -3
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  panel %.3f  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['panel'], j['result']['logpdf']))"
}
for round in 1 2 3; do
  for v in 0 1; do echo "round $round C4 SMN_PANEL_HELPERS=$v"; SMN_PANEL_HELPERS=$v one --steps 20 --warmup 3; done
done
for v in 0 1; do echo "C2 SMN_PANEL_HELPERS=$v"; SMN_PANEL_HELPERS=$v one --config c2 --steps 50 --warmup 5; done
for v in 0 1; do echo "f64 n8192 SMN_PANEL_HELPERS=$v"; SMN_PANEL_HELPERS=$v one --dtype f64 --n 8192 --steps 10 --warmup 2; done
for v in 0 1; do echo "C5 SMN_PANEL_HELPERS=$v"; SMN_PANEL_HELPERS=$v one --config c5 --steps 4 --warmup 1; done
