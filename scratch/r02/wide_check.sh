#!/bin/bash
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0)))"
}
for n in 20480 24576 28672; do for r in 1099511627776 18432; do echo "N=$n d=1024 SMN_SUPER_WIDE_ROWS=$r"; SMN_SUPER_WIDE_ROWS=$r one --n $n --d 1024 --steps 4 --warmup 1; done; done
echo "C5 default"; one --config c5 --steps 4 --warmup 1
echo "C4 default"; one --steps 20 --warmup 3
timeout -k 10 600 python3 -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -2
