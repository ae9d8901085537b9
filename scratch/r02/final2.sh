#!/bin/bash
# r02 evidence batch after the tile-engine change: final_measure.sh, then the MFMA-busy counter passes (two groups only).
set -u
R=$GRAFT_REPO_ROOT
bash $R/scratch/r02/final_measure.sh
OUT=$R/gpurun_out/r02_final/pmc_busy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SMN_CHAIN_CUS=0
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $OUT/p$i.log 2>&1
  echo "== $grp (rc=$?)"
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/p$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        g = int(r["Grid_Size"]) // 256 if "Grid_Size" in r else 0
        if k.startswith("update_kernel<float, 1, 128, 128>") and g < 3000: continue   # only the big far updates
        if k.startswith(("update_kernel<float, 1, 128, 128>", "build_kernel", "trail_kernel")):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    print("  ", k[:40], {c: "%.4g" % v for c, v in agg[k].items()})
PY
done
