#!/bin/bash
# How does the far update's throughput scale with the CUs it may use?  (SMN_CHAIN_CUS = CUs it may NOT use; n = 24576 keeps it short.)
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f  trail(sum) %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], j['phases_ms']['trail']))"
}
for c in 32 64 96 128 160 192; do echo "N=24576 d=1024 SMN_CHAIN_CUS=$c"; SMN_CHAIN_CUS=$c one --n 24576 --d 1024 --steps 3 --warmup 1; done
