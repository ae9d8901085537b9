#!/bin/bash
# r03: look-ahead (F0/F1 split on the CU-masked bulk stream) at C2's size: threshold x super-panel width
for cfg in "8192 1024" "4096 1024" "4096 512" "4096 256" "2048 512" "2048 256"; do
  set -- $cfg
  for f0 in 2000 0; do
    r=$(SMN_CHAIN_MIN_N=$1 SMN_SUPER=$2 SMN_F0_FIRST_TILES=$f0 python3 bench.py --config c2 --steps 200 --warmup 20 --no-other-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
    echo "chain_min_n $1 super $2 f0_first_tiles $f0: c2 $r ms"
  done
done
