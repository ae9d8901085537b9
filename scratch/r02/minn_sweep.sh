#!/bin/bash
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0)))"
}
for n in 6144 8192 10240 12288; do for m in 1 1000000; do echo "N=$n d=1024 SMN_CHAIN_MIN_N=$m ($([ $m = 1 ] && echo look-ahead on || echo off))"; SMN_CHAIN_MIN_N=$m one --n $n --d 1024 --steps 20 --warmup 3; done; done
for n in 8192 12288; do for s in 512 1024; do echo "N=$n d=1024 look-ahead on SMN_SUPER=$s"; SMN_CHAIN_MIN_N=1 SMN_SUPER=$s one --n $n --d 1024 --steps 20 --warmup 3; done; done
