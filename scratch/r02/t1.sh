#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t1.log 2>&1
echo "pytest rc=$?"; tail -15 $O/t1.log
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-recursion-probe > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['phases_ms'],d['roofline']['cholesky_wall_ms'],d['roofline']['frac'],d['roofline'].get('frac_exclusive'), d['result'])"
tail -3 $O/bench.err
