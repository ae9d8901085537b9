#!/bin/bash
# The XCD patch order, now compact and balanced (TileMap): correctness, A/B against the linear order, fabric traffic of the
# trailing updates with it (FETCH_SIZE / WRITE_SIZE, one pass each).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_xcd2
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "schedule_variants or gram_then_recursion or symmetric_kernel" 2>&1 | tail -3
one() {
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f  frac_excl %s  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], r.get('frac_exclusive'), j['result']['logpdf']))"
}
for round in 1 2 3; do
  for v in 0 1; do echo "round $round C4 SMN_XCD_MAP=$v"; SMN_XCD_MAP=$v one --steps 20 --warmup 3; done
done
for v in 0 1; do echo "C5 SMN_XCD_MAP=$v"; SMN_XCD_MAP=$v one --config c5 --steps 4 --warmup 1; done
cd /tmp && export TMPDIR=/tmp
export SMN_XCD_MAP=1
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/c4_$grp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/c4_$grp.log 2>&1
  echo "c4 $grp rc=$?"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/c4_%s/*/*counter_collection.csv" % grp):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ("update_kernel<float, 1" in k) or ("trail_kernel" in k):
                tot[grp] += float(r["Counter_Value"]); cnt[grp] += 1
print("C4 trailing launches (SMN_XCD_MAP=1):", dict(cnt), {k: "%.4g KB" % v for k, v in tot.items()})
if cnt.get("FETCH_SIZE"):
    per = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / cnt["FETCH_SIZE"]
    print("bytes per launch (FETCH x 2 + WRITE): %.1f MB" % (per / 1e6))
PY
