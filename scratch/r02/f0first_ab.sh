#!/bin/bash
# F1 behind F0 (instead of beside it) once F1 has at most T tiles: sweep of T.
set -e
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['result']['logpdf']))"
}
for round in 1 2; do
  for t in 0 400 1000 1800 2800 4000 100000; do echo "round $round C4 SMN_F0_FIRST_TILES=$t"; SMN_F0_FIRST_TILES=$t one --steps 20 --warmup 3; done
done
for t in 0 1800 4000; do echo "C5 SMN_F0_FIRST_TILES=$t"; SMN_F0_FIRST_TILES=$t one --config c5 --steps 4 --warmup 1; done
