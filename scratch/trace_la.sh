#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_la.so SMN_LOOKAHEAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_la -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe > $R/gpurun_out/trace_la.json 2> $R/gpurun_out/trace_la.err
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_la/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0], r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
# take the last step: find last build_kernel
idx = max(i for i, r in enumerate(rows) if r[2].startswith("build_kernel"))
t0 = rows[idx][0]
sel = [r for r in rows[idx:idx+40]]
for s, e, n, q in sel:
    print("%9.1f %9.1f  %-28s q=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:28], q))
PY
