#!/bin/bash
# software-pipelined K loop of the tile engine (default) against the plain loop (variant build -DSMN_PIPE=0)
run() { echo "== $1 $2"; SMNNGP_LIB=$PWD/scale-mixtures-of-neural-network-gaussian-processes_amd/$1 timeout -k 10 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-recursion-probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(round(d['ms_per_step'],3), d['phases_ms'], 'excl', r.get('frac_exclusive'), d['result']['logpdf'])
    elif l: print(l[:300])"; }
for rep in 1 2; do
run libsmnngp.so ""
run libsmnngp_nopipe.so ""
done
run libsmnngp.so "--n 4096 --d 512 --layers 3"
run libsmnngp_nopipe.so "--n 4096 --d 512 --layers 3"
run libsmnngp.so "--dtype f64 --n 8192"
run libsmnngp_nopipe.so "--dtype f64 --n 8192"
run libsmnngp.so "--n 32768 --d 1024 --layers 6 --act erf"
run libsmnngp_nopipe.so "--n 32768 --d 1024 --layers 6 --act erf"
