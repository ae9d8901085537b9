#!/bin/bash
# Wide (2048-column) super-panels while at least R rows are left, 1024 below: sweep of R.
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['result']['logpdf']))"
}
timeout -k 10 300 env SMN_SUPER_WIDE_ROWS=4096 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or lml or predict" 2>&1 | tail -2
for round in 1 2; do
for r in 1099511627776 15000 13000 11000 9000 7000; do echo "round $round C4 SMN_SUPER_WIDE_ROWS=$r"; SMN_SUPER_WIDE_ROWS=$r one --steps 20 --warmup 3; done
done
for r in 1099511627776 30000 26000 22000 18000 14000 10000; do echo "C5 SMN_SUPER_WIDE_ROWS=$r"; SMN_SUPER_WIDE_ROWS=$r one --config c5 --steps 4 --warmup 1; done
