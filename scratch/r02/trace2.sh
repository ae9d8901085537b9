#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02c}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
cd $R
python3 scratch/trace_analyze.py $O/trace > $O/timeline.txt 2>&1
tail -50 $O/timeline.txt
find $O/trace -name "*kernel_stats.csv" -exec head -14 {} \; | cut -c1-180
rm -f $O/trace/*/*_kernel_trace.csv.bak
