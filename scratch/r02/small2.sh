#!/bin/bash
set -e
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python3 scratch/small_n_latency.py
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 %.3f ms/step' % j['ms_per_step'], j['result'])"
