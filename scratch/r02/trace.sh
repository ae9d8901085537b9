#!/bin/bash
# r02 baseline: default bench line + a kernel trace (timeline) of two steps
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
cd $R
python3 scratch/trace_analyze.py $O/trace > $O/timeline.txt 2>&1
cat $O/timeline.txt | tail -60
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['phases_ms'],d['roofline']['cholesky_wall_ms'],d['roofline']['frac'],d['roofline'].get('frac_exclusive'))"
