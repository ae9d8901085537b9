#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_rec2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PL=8 timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/p1 -- python3 $R/scratch/rec_probe.py > $OUT/p1.log 2> $OUT/p1.err
cd $R; tail -1 $OUT/p1.log; python3 scratch/pmc_summary.py $OUT | grep recursion | cut -c1-900
