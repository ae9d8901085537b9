#!/bin/bash
# C5 shape (N = 32768): how many CUs should the panel chain keep?  (C4 was swept in r01/r02: flat between 16 and 48.)
set -e
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build']))"
}
for c in 32 8 16 24 48; do echo "C5 SMN_CHAIN_CUS=$c"; SMN_CHAIN_CUS=$c one --config c5 --steps 4 --warmup 1; done
for c in 32 16 24; do echo "C4 SMN_CHAIN_CUS=$c"; SMN_CHAIN_CUS=$c one --steps 20 --warmup 3; done
