#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['phases_ms']['trail'], d['phases_ms']['panel'])
    elif l: print(l[:300])"; }
run SMN_CHAIN_CUS=16
run SMN_CHAIN_CUS=24
run SMN_CHAIN_CUS=32
run SMN_CHAIN_CUS=40
run SMN_CHAIN_CUS=24
run SMN_CHAIN_CUS=32
