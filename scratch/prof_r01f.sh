#!/bin/bash
# Round-1e evidence: default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01f
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/r01f/bench.json 2> $R/gpurun_out/r01f/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01f/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $R/gpurun_out/r01f/stats.json 2> $R/gpurun_out/r01f/stats.err
cd $R
bash scratch/pmc.sh r01f FETCH_SIZE WRITE_SIZE > $R/gpurun_out/r01f/pmc.txt 2>&1
tail -c 1500 $R/gpurun_out/r01f/bench.json
find $R/gpurun_out/r01f/stats -name "*kernel_stats.csv" -exec head -8 {} \;
cat $R/gpurun_out/r01f/pmc.txt
