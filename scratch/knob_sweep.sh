#!/bin/bash
# Joint re-tune of the schedule knobs on the current build (C4 shape); each setting run twice, interleaved.
run() { env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.3f' % d['ms_per_step'], '$*')"; }
for rep in 1 2; do
run SMN_NOP=1
run SMN_CHAIN_CUS=24
run SMN_CHAIN_CUS=40
run SMN_HALF_TILES=512
run SMN_HALF_TILES=768
run SMN_HALF_TILES=256
run SMN_QUARTER_TILES=128
run SMN_QUARTER_TILES=384 SMN_HALF_TILES=512
run SMN_PANEL_SMALL=2048
run SMN_PANEL_SMALL=8192
run SMN_PANEL_SMALL=0
run SMN_PERSIST_MAXK=256
run SMN_SUPER=1536
run SMN_SUPER=768
run SMN_CHAIN_MIN_N=100000
done
