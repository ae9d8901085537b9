#!/bin/bash
# After the engine / panel / order changes of this round: the tile-shape and panel-size thresholds once more (C4, C2).
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0)))"
}
echo "C4 default"; one --steps 20 --warmup 3
for kv in SMN_HALF_TILES=256 SMN_HALF_TILES=512 SMN_HALF_TILES=768 SMN_QUARTER_TILES=128 SMN_QUARTER_TILES=384 SMN_QUARTER_TILES=512 SMN_PANEL_SMALL=2048 SMN_PANEL_SMALL=8192 SMN_PANEL_SMALL=16384 SMN_PERSIST_MAXK=256 SMN_PERSIST_MAXK=1024 SMN_PERSISTENT=0; do
  echo "C4 $kv"; env $kv bash -c "$(declare -f one); one --steps 20 --warmup 3"
done
echo "C4 default again"; one --steps 20 --warmup 3
echo "C2 default"; one --config c2 --steps 50 --warmup 5
for kv in SMN_QUARTER_TILES=512 SMN_QUARTER_TILES=1024 SMN_HALF_TILES=1024 SMN_PANEL_SMALL=0 SMN_PERSISTENT=0; do
  echo "C2 $kv"; env $kv bash -c "$(declare -f one); one --config c2 --steps 50 --warmup 5"
done
