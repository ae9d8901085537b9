#!/bin/bash
# r02 measurement batch: (1) MFMA peak microbench, (2) VALU-busy / wait counters of the final conv_pair32_kernel<double> at
# N = 2048 (C3's kernel), (3) FETCH_SIZE / WRITE_SIZE of one C4 bench step (all trailing-update launches).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_misc
mkdir -p $O
cd $R
./scratch/r02/mfma_peak/mfma_peak > $O/mfma_peak.txt 2>&1; cat $O/mfma_peak.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/cnn_p$i -- python3 $R/scratch/cnn_one.py 2048 f64 > $O/cnn_p$i.log 2>&1
  echo "cnn pass $i ($grp) rc=$?"
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/c4_$grp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/c4_$grp.log 2>&1
  echo "c4 $grp rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(float); n = collections.Counter(); dur = []
for f in glob.glob("$O/cnn_p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob("$O/cnn_p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_pair" in r["Kernel_Name"]:
            dur.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:60]))
per = {k: agg[k] / max(n[k], 1) for k in agg}
print("conv kernel launches:", dur)
for k in sorted(per): print("   %-24s %.5g" % (k, per[k]))
out = {"kernel": dur[0][1] if dur else None, "N": 2048, "counters_per_launch": per}
if "SQ_ACTIVE_INST_VALU" in per and "SQ_BUSY_CYCLES" in per:
    pass
# VALU busy = SQ_ACTIVE_INST_VALU * 4 / (SIMDs) / busy cycles: the counter counts cycles a SIMD's VALU executes, summed over SIMDs
if "SQ_ACTIVE_INST_VALU" in per and "SQ_WAVE_CYCLES" in per:
    out["valu_active_over_wave_cycles"] = per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"]
if "SQ_WAIT_INST_ANY" in per and "SQ_WAVE_CYCLES" in per:
    out["wait_inst_over_wave_cycles"] = per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"]
json.dump(out, open("$O/cnn_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters_per_launch"}))
# C4 traffic: every trailing-update launch of the (one) warm-up + timed + detail steps
tot = collections.defaultdict(float); cnt = collections.Counter()
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/c4_%s/*/*counter_collection.csv" % grp):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ("update_kernel<float, 1" in k) or ("trail_kernel" in k):
                tot[grp] += float(r["Counter_Value"]); cnt[grp] += 1
print("C4 trailing launches:", dict(cnt), {k: "%.4g KB" % v for k, v in tot.items()})
json.dump({"FETCH_SIZE_KB": tot.get("FETCH_SIZE"), "WRITE_SIZE_KB": tot.get("WRITE_SIZE"), "launches": cnt.get("FETCH_SIZE")}, open("$O/c4_traffic_raw.json", "w"))
PY
