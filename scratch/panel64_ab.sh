#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], d['phases_ms']['panel'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
run SMN_PANEL_SMALL=0
run SMN_PANEL_SMALL=4096
run SMN_PANEL_SMALL=0
run SMN_PANEL_SMALL=4096
run SMN_PANEL_SMALL=2048
run SMN_PANEL_SMALL=6144
