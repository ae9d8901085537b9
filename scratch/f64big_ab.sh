#!/bin/bash
# f64 128x128 tile with the pipelined K loop (variant build -DSMN_PIPE_F64_BIG=1: spills) against the plain loop
run() { echo "== $1"; SMNNGP_LIB=$PWD/scale-mixtures-of-neural-network-gaussian-processes_amd/$1 timeout -k 10 300 python bench.py --dtype f64 --n 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-recursion-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(round(d['ms_per_step'],3), d['phases_ms'], 'excl', r.get('frac_exclusive'), d['result']['logpdf'])
    elif l: print(l[:300])"; }
for rep in 1 2; do run libsmnngp.so; run libsmnngp_f64big.so; done
