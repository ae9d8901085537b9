#!/bin/bash
# Counter passes over the conv-NNGP pair kernel (fp64 and fp32), one counter group per run.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_cnn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for dt in f64 f32; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/scratch/cnn_one.py 1024 $dt > $OUT/p$i.log 2>&1
  echo "== $dt :: $grp (rc=$?)"
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/p$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print("   %-24s %.4g  (per launch, %d launches)" % (k, agg[k] / max(n[k], 1), n[k]))
PY
done
done
