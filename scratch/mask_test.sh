#!/bin/bash
# Does hipExtStreamCreateWithCUMask restrict kernels here?  Bench phases with the main stream on n CUs.
for n in 256 128 64; do
  echo "== SMN_MAIN_MASK_CUS=$n"
  SMN_MAIN_MASK_CUS=$n timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recursion-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['phases_ms'])
    elif l: print(l[:300])"
done
