#!/bin/bash
# r03: super-panel width at C2's size (look-ahead stays off below N = 8192)
for s in 256 512 1024 2048 4096; do
  r=$(SMN_SUPER=$s python3 bench.py --config c2 --steps 300 --warmup 30 --no-other-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
  echo "super $s: c2 $r ms"
done
for p in 0 1; do
  r=$(SMN_PERSISTENT=$p python3 bench.py --config c2 --steps 300 --warmup 30 --no-other-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
  echo "persistent $p: c2 $r ms"
done
