#!/bin/bash
# F0 split (first outer panel's columns on the chain's stream, the rest on a second masked stream) on / off
run() { echo "== $1 :: $2"; env $1 timeout -k 10 300 python bench.py $2 --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],3), d['phases_ms'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
for rep in 1 2 3; do
run SMN_F0_SPLIT=1 ""
run SMN_F0_SPLIT=0 ""
done
run SMN_F0_SPLIT=1 "--n 32768 --d 1024 --layers 6 --act erf"
run SMN_F0_SPLIT=0 "--n 32768 --d 1024 --layers 6 --act erf"
run SMN_F0_SPLIT=1 "--dtype f64 --n 8192"
run SMN_F0_SPLIT=0 "--dtype f64 --n 8192"
