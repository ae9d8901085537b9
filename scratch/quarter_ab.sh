#!/bin/bash
# A/B of the 64x64 quarter-tile updates (SMN_QUARTER_TILES = largest 128x128-tile count that takes them)
run() { echo "== $*"; env "${@:2}" timeout -k 10 300 python bench.py $1 --steps 10 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], d['phases_ms'], d['result']['logpdf'])
    elif l: print(l[:300])"; }
for q in 0 64 128 192 256 384 0 128; do run "" SMN_QUARTER_TILES=$q; done
for q in 0 64 128 256 512; do run "--n 4096 --d 512 --layers 3" SMN_QUARTER_TILES=$q; done
for q in 0 128; do run "--dtype f64 --n 8192" SMN_QUARTER_TILES=$q; done
