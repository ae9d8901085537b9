#!/bin/bash
# persistent near-update kernel (plain K loop inside) against the pipelined update_kernel for the same launches
run() { echo "== $1 :: $2"; env $1 timeout -k 10 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],3), d['phases_ms'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
for rep in 1 2; do
run SMN_PERSISTENT=1 ""
run SMN_PERSISTENT=0 ""
done
run SMN_PERSISTENT=1 "--n 32768 --d 1024 --layers 6 --act erf"
run SMN_PERSISTENT=0 "--n 32768 --d 1024 --layers 6 --act erf"
