#!/bin/bash
# Kernel sequence (names, durations, gaps) of one reference-sized SPR.loss (N=245) under rocprofv3 --kernel-trace, and latencies.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_small
mkdir -p $O
cd $R && timeout -k 10 200 python3 scratch/small_n_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scratch/small_trace.py > $O/trace.log 2>&1; echo "trace rc=$?"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/trace/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the third loss call: find sequences starting at pad_rows
names = [r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:48] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("pad_rows")]
# loss = two pad_rows in a row at its head; print the kernels of the 3rd loss call
heads = [i for k, i in enumerate(starts) if k % 2 == 0]
i0, i1 = heads[2], heads[3]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for i in range(i0, i1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  +%5.1f gap  %6.1f us  %s  grid %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, names[i], rows[i].get("Grid_Size")))
    prev_end = e
print("span %.1f us" % ((prev_end - t0) / 1e3))
PY
