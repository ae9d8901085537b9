#!/bin/bash
# r03: counter passes over the conv-NNGP pair kernel (conv_pair44_kernel<double,0,3>), one counter group per run, N = 2048.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_pmc_cnn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/scratch/cnn_one.py 2048 f64 > $OUT/p$i.log 2>&1
  echo "== $grp (rc=$?)"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(float); n = collections.Counter(); dur = []
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob("$OUT/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_pair" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6); name = r["Kernel_Name"]
c = {k: agg[k] / n[k] for k in agg}
quads = c["GRBM_GUI_ACTIVE"] / 8 * 1024 / 4
out = {"workload": "smn_kernel_cnn, N=2048 images 32x32x3, 4-layer ReLU, fp64: scratch/cnn_one.py 2048 f64", "kernel": name,
       "ms_per_launch": sum(dur) / len(dur), "counters_per_launch": c,
       "valu_busy": c["SQ_ACTIVE_INST_VALU"] / quads, "valu_instr_floor": c["SQ_INSTS_VALU"] / quads,
       "wave_wait_any_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "wave_wait_on_instruction_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
       "resident_waves_per_simd": c["SQ_WAVE_CYCLES"] / quads,
       "valu_instructions_per_pair_pixel_layer": c["SQ_INSTS_VALU"] * 64 / (2048 * 2049 / 2 * 1024 * 4),
       "valu_busy_definition": "SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs / 4): share of all SIMD issue quads in which a VALU instruction executes",
       "source": "profiles/r03_pmc_cnn.json: rocprofv3 --kernel-trace --pmc, one counter group per run (scratch/r03/pmc_cnn.sh)"}
json.dump(out, open("$OUT/r03_pmc_cnn.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "counters_per_launch"}, indent=1))
PY
