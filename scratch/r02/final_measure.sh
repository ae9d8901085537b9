#!/bin/bash
# r02 evidence batch: full GPU test suite, the default bench line, rocprofv3 kernel stats of the same command, the other
# configurations, the one-rank rehearsal of the sharded route, a look-ahead A/B at C2.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_final
mkdir -p $O
rm -rf $O/stats
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/stats.json 2> $O/stats.err; echo "stats rc=$?"
cd $R
for cfg in c2 c5 c3; do
  timeout -k 10 500 python3 bench.py --config $cfg --steps 10 --warmup 2 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"
done
timeout -k 10 300 python3 bench.py --dtype f64 --n 8192 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_f64_n8192.json 2> $O/bench_f64.err; echo "f64 rc=$?"
timeout -k 10 300 python3 bench.py --sharded-path --no-cpu-baseline > $O/bench_sharded_one_rank.json 2> $O/bench_sharded.err; echo "sharded rc=$?"
for mn in 8192 2048; do echo -n "c2 SMN_CHAIN_MIN_N=$mn: "; SMN_CHAIN_MIN_N=$mn timeout -k 10 100 python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('%.3f ms/step'%d['ms_per_step'])"; done | tee $O/c2_lookahead_ab.txt
python3 - <<PY
import json
for f in ("bench", "bench_c2", "bench_c5", "bench_c3", "bench_f64_n8192", "bench_sharded_one_rank"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.3f ms/step" % d["ms_per_step"], "value %.4g %s" % (d["value"], d["unit"]), d.get("phases_ms"), "chol", d["roofline"].get("cholesky_wall_ms"), "frac", d["roofline"].get("frac"), d["roofline"].get("frac_exclusive"), "cpu", (d.get("cpu_baseline") or {}).get("rel_diff_vs_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
find $O/stats -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-170
