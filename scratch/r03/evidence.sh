#!/bin/bash
# r03 evidence batch: full GPU test suite, the default bench line (with other_workloads), rocprofv3 kernel stats of the same
# command, PMC traffic of the trailing update, panel timeline (timing build), recursion table, sharded one-rank rehearsal.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
rm -rf $O/stats $O/pmc
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads > $O/stats.json 2> $O/stats.err; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
done
cd $R
SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_timing.so timeout -k 10 120 python3 scratch/r03/panel_timing.py > $O/panel_timeline.txt 2>&1; echo "timeline rc=$?"
timeout -k 10 300 bash scratch/rec_table.sh > $O/recursion_table.txt 2>&1; echo "rec table rc=$?"
timeout -k 10 200 python3 bench.py --sharded-path --no-cpu-baseline > $O/bench_sharded_one_rank.json 2> $O/bench_sharded.err; echo "sharded rc=$?"
python3 - <<PY
import csv, glob, json, collections
d = json.load(open("$O/bench.json"))
print("C4 %.3f ms/step  %.1f GFLOP/s  frac %.3f  excl %.3f  chol %.3f  phases %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("frac_exclusive") or 0, d["roofline"]["cholesky_wall_ms"], d["phases_ms"]))
for k, v in d.get("other_workloads", {}).items():
    print("  ", k, v.get("ms_per_step") or {kk: vv.get("ms_per_call") for kk, vv in v.items() if isinstance(vv, dict) and "ms_per_call" in vv}, v.get("error") or v.get("skipped") or "")
agg = collections.defaultdict(float); n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/pmc/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "update_kernel" in r["Kernel_Name"] and ", 1" in r["Kernel_Name"] or "trail_kernel" in r["Kernel_Name"]:
                agg[c] += float(r["Counter_Value"]); n[c] += 1
print("trailing update PMC:", dict(agg), dict(n))
if n["FETCH_SIZE"]:
    per = (2 * agg["FETCH_SIZE"] * 1024 / n["FETCH_SIZE"]) + agg["WRITE_SIZE"] * 1024 / max(n["WRITE_SIZE"], 1)
    print("traffic bytes per launch (FETCH x2 correction + WRITE): %.4g over %d launches" % (per, n["FETCH_SIZE"]))
    json.dump({"launches": n["FETCH_SIZE"], "FETCH_SIZE_KB": agg["FETCH_SIZE"], "WRITE_SIZE_KB": agg["WRITE_SIZE"], "traffic_bytes_per_launch": per}, open("$O/pmc_traffic_raw.json", "w"))
PY
find $O/stats -name "*kernel_stats.csv" -exec head -14 {} \; | cut -c1-200
