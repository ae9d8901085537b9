#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recursion-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], d['phases_ms']['trail'], d['phases_ms']['build'], r['frac'], r.get('frac_exclusive'))
    elif l: print(l[:300])"; }
run A=1
run SMN_XCD_MAP=1
run A=2
run SMN_XCD_MAP=1
