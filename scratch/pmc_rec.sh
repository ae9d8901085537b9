#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_rec; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/scratch/rec_probe.py > $OUT/p$i.log 2> $OUT/p$i.err
done
cd $R; tail -1 $OUT/p1.log; python3 scratch/pmc_summary.py $OUT | grep recursion | cut -c1-1200
