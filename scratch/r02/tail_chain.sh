#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_tail
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
python3 $R/scratch/r02/tail_chain.py $O/trace
