#!/bin/bash
# Far updates at ONE workgroup per CU (96 KB of LDS asked for) against two: the tile probe saw 83.5 % against 78.5 %.
set -e
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f  frac_excl %s  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], r.get('frac_exclusive'), j['result']['logpdf']))"
}
for round in 1 2; do
  echo "round $round default"; one --steps 20 --warmup 3
  echo "round $round PERSIST_MAXK=0 (no persistent kernel anywhere)"; SMN_PERSIST_MAXK=0 one --steps 20 --warmup 3
  echo "round $round 96 KB"; SMN_DEBUG_UPD_LDS_KB=96 SMN_PERSIST_MAXK=0 one --steps 20 --warmup 3
done
echo "C5 default"; one --config c5 --steps 4 --warmup 1
echo "C5 96 KB"; SMN_DEBUG_UPD_LDS_KB=96 one --config c5 --steps 4 --warmup 1
