#!/bin/bash
# PMC passes over one bench step (each pass its own run; --pmc only with --kernel-trace).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $R
python3 scratch/pmc_summary.py $OUT
