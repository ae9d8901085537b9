#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --dtype f64 --n 8192 --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['phases_ms'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
run SMN_CHAIN_CUS=0
run SMN_CHAIN_CUS=32
run SMN_CHAIN_CUS=64
run SMN_CHAIN_CUS=128
