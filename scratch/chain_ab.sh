#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['phases_ms'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
run SMN_LOOKAHEAD=1 SMN_CHAIN_CUS=16
run SMN_LOOKAHEAD=1 SMN_CHAIN_CUS=32
run SMN_LOOKAHEAD=1 SMN_CHAIN_CUS=64
run SMN_LOOKAHEAD=1 SMN_CHAIN_CUS=32 SMN_PERSISTENT=0
run SMN_MAIN_PRIO_HI=1 SMN_CHAIN_CUS=24
run SMN_MAIN_PRIO_HI=1 SMN_CHAIN_CUS=40
run SMN_MAIN_PRIO_HI=1 SMN_CHAIN_CUS=32 SMN_SUPER=2048
run SMN_MAIN_PRIO_HI=1 SMN_CHAIN_CUS=32 SMN_SUPER=512
