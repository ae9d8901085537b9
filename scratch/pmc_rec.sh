#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) + kernel trace over the stand-alone recursion probe.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_rec
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/scratch/rec_probe.py > $OUT/p$i.out 2> $OUT/p$i.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scratch/rec_probe.py > $OUT/stats.out 2> $OUT/stats.err
cd $R
python3 scratch/pmc_summary.py $OUT
cat $OUT/summary.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cat {} \;
