#!/bin/bash
# one round of the panelf_kernel work: bit-identity, kernel durations under rocprofv3, timeline of the timing build
R=$PWD
O=$R/gpurun_out/r04c
mkdir -p $O
timeout -k 10 120 python3 $R/scratch/r04/panelf_check.py quick > $O/check.txt 2>&1 || { cat $O/check.txt; echo "check failed"; exit 1; }
tail -1 $O/check.txt
cd /tmp && export TMPDIR=/tmp
for leaf in 2 1; do
  rm -rf $O/k$leaf
  SMN_PANEL_LEAF=$leaf timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$leaf -- python3 $R/scratch/r04/panelf_one.py > $O/k$leaf.log 2>&1 || { echo "trace $leaf failed"; exit 1; }
  python3 - $O/k$leaf <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in sorted(csv.reader(open(f))):
    if "panel" in r[0]:
        print("  ", r[0].split("::")[-1][:34], "calls", r[1], "avg %.1f us  min %.1f us" % (float(r[3]) / 1e3, float(r[5]) / 1e3))
PY
done
cd $R
SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_timing.so SMN_PANEL_LEAF=2 timeout -k 10 120 python3 scratch/panel_timing.py 2048 8192 > $O/timeline_f.txt 2>&1
echo timeline rc=$?
