#!/bin/bash
# kernel durations of panelf_kernel A/B builds (rocprofv3 --kernel-trace --stats of scratch/r04/panelf_one.py)
R=$PWD
mkdir -p $R/gpurun_out/r04c
cd /tmp && export TMPDIR=/tmp
for v in "" _pfu _pfn _pfun; do
  export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp$v.so
  SMN_PANEL_LEAF=2 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04c/v$v -- python3 $R/scratch/r04/panelf_one.py > $R/gpurun_out/r04c/v$v.log 2>&1 || { echo "variant $v failed"; exit 1; }
  echo "variant '$v'"
  python3 - $R/gpurun_out/r04c/v$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.reader(open(f)):
    if "panel" in r[0]:
        print("  ", r[0].split("::")[-1][:34], "calls", r[1], "avg %.1f us  min %.1f us" % (float(r[3]) / 1e3, float(r[5]) / 1e3))
PY
done
