#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_noev
SMN_BENCH_NOPROF=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_noev -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $R/gpurun_out/trace_noev.json 2> $R/gpurun_out/trace_noev.err
cd $R
python3 scratch/trace_analyze.py gpurun_out/trace_noev
