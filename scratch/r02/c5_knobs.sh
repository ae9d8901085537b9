#!/bin/bash
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0)))"
}
echo "C5 default"; one --config c5 --steps 4 --warmup 1
for kv in SMN_CHAIN_CUS=16 SMN_CHAIN_CUS=24 SMN_CHAIN_CUS=40 SMN_SUPER_WIDE=3072 SMN_SUPER_WIDE=4096 SMN_F0_FIRST_TILES=8000 SMN_F0_FIRST_TILES=30000 SMN_PERSIST_MAXK=2048 SMN_XCD_MAP=0; do
  echo "C5 $kv"; env $kv bash -c "$(declare -f one); one --config c5 --steps 4 --warmup 1"
done
echo "C5 default again"; one --config c5 --steps 4 --warmup 1
