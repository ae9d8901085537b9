#!/bin/bash
# r04 evidence batch (one gpurun call): full GPU test suite, the default bench line (with other_workloads), rocprofv3 kernel
# stats of the same command, PMC traffic of the trailing update and of the stand-alone recursion, recursion table, the
# one-rank rehearsal of the multi-GPU step, C5 with its NTK through the same route.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_final
mkdir -p $O
rm -rf $O/stats $O/pmc $O/pmc_rec
cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads > $O/stats.json 2> $O/stats.err; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_rec/$c -- python3 $R/scratch/rec_probe.py > $O/pmc_rec_$c.out 2> $O/pmc_rec_$c.err; echo "pmc rec $c rc=$?"
done
cd $R
timeout -k 10 200 python3 scratch/r04/recursion_table.py > $O/recursion_table.txt 2>&1; echo "rec table rc=$?"
timeout -k 10 200 python3 bench.py --sharded-path --no-cpu-baseline > $O/bench_sharded_one_rank.json 2> $O/bench_sharded.err; echo "sharded rc=$?"
timeout -k 10 300 python3 bench.py --sharded-path --config c5 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_sharded_one_rank_c5_ntk.json 2> $O/bench_sharded_c5.err; echo "sharded c5 rc=$?"
python3 - <<PY
import csv, glob, json, collections
d = json.load(open("$O/bench.json"))
print("C4 %.3f ms/step  %.1f GFLOP/s  frac %.3f  excl %.3f  chol %.3f  phases %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("frac_exclusive") or 0, d["roofline"]["cholesky_wall_ms"], d["phases_ms"]))
for k, v in d.get("other_workloads", {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ms_per_sweep", "device_batches_ms", "serial_calls_ms", "speedup_vs_serial_calls", "loss_evaluations", "batched_us_per_problem", "serial_us_per_problem")} or {kk: vv.get("ms_per_call") for kk, vv in v.items() if isinstance(vv, dict) and "ms_per_call" in vv}, v.get("error") or v.get("skipped") or "")
def pmc(root, match):
    agg = collections.defaultdict(float); n = collections.Counter()
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s/*/*counter_collection.csv" % (root, c)):
            for r in csv.DictReader(open(f)):
                if match(r["Kernel_Name"]):
                    agg[c] += float(r["Counter_Value"]); n[c] += 1
    if not n["FETCH_SIZE"]:
        return None
    per = (2 * agg["FETCH_SIZE"] * 1024 / n["FETCH_SIZE"]) + agg["WRITE_SIZE"] * 1024 / max(n["WRITE_SIZE"], 1)
    return {"launches": n["FETCH_SIZE"], "FETCH_SIZE_KB": agg["FETCH_SIZE"], "WRITE_SIZE_KB": agg["WRITE_SIZE"], "traffic_bytes_per_launch": per}
t = pmc("$O/pmc", lambda k: ("update_kernel" in k and ", 1" in k) or "trail_kernel" in k)
print("trailing update PMC:", t)
json.dump(t, open("$O/pmc_traffic_raw.json", "w"))
r = pmc("$O/pmc_rec", lambda k: "recursion_sym_kernel" in k)
print("recursion PMC:", r)
json.dump(r, open("$O/pmc_recursion_raw.json", "w"))
PY
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -14 $O/kernel_stats.csv | cut -c1-220
rm -rf $O/stats $O/pmc $O/pmc_rec
