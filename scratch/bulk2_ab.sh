#!/bin/bash
# adaptive CU reservation: late far updates (<= T tiles) on a stream that leaves R CUs to the chain
run() { env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.3f' % d['ms_per_step'], '$*', d['result']['logpdf'])"; }
for rep in 1 2; do
run SMN_BULK2_TILES=0
run SMN_BULK2_TILES=1700 SMN_CHAIN_CUS2=64
run SMN_BULK2_TILES=1700 SMN_CHAIN_CUS2=96
run SMN_BULK2_TILES=2800 SMN_CHAIN_CUS2=64
run SMN_BULK2_TILES=2800 SMN_CHAIN_CUS2=96
run SMN_BULK2_TILES=900 SMN_CHAIN_CUS2=96
run SMN_BULK2_TILES=900 SMN_CHAIN_CUS2=128
run SMN_BULK2_TILES=4100 SMN_CHAIN_CUS2=48
done
