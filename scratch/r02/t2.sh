#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02g}
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -8 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['phases_ms'],d['roofline']['cholesky_wall_ms'],d['roofline']['frac'],d['roofline'].get('frac_exclusive'), d['result']); print(d['cpu_baseline'])"
tail -3 $O/bench.err
