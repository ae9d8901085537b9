#!/bin/bash
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build']))"
}
for n in 20480 24576 28672; do for s in 1024 2048; do echo "N=$n d=1024 SMN_SUPER=$s"; SMN_SUPER=$s one --n $n --d 1024 --steps 4 --warmup 1; done; done
for s in 3072 4096; do echo "C5 SMN_SUPER=$s"; SMN_SUPER=$s one --config c5 --steps 4 --warmup 1; done
