#!/bin/bash
# L2 behaviour of the big far updates (look-ahead off: nothing else on the GPU), linear tile order vs XCD patch order.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_pmc_l2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SMN_CHAIN_MIN_N=1000000000
for MAP in 0 1; do
  i=0
  for grp in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    i=$((i+1))
    SMN_XCD_MAP=$MAP timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/m${MAP}_p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/m${MAP}_p$i.log 2>&1
    echo "map=$MAP pass $i ($grp) rc=$?"
  done
done
cd $R
python3 - <<PY
import csv, glob, collections, json
out = {}
for MAP in (0, 1):
    per = collections.defaultdict(dict)   # (kernel, grid) -> counter -> list of values per dispatch
    durs = collections.defaultdict(list)
    for p in range(1, 6):
        for f in glob.glob("$O/m%d_p%d/*/*counter_collection.csv" % (MAP, p)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
                if not k.startswith(("update_kernel<float, 1, 128, 128>", "build_kernel")): continue
                g = int(r["Grid_Size"]) // 256
                per[(k, g)].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in glob.glob("$O/m%d_p%d/*/*kernel_trace.csv" % (MAP, p)):
            if p != 1: continue
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
                g = int(r["Grid_Size_X"]) // 256
                durs[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for (k, g), c in sorted(per.items(), key=lambda kv: -kv[0][1])[:14]:
        m = {n: sum(v) / len(v) for n, v in c.items()}
        m["us"] = sum(durs.get((k, g), [0])) / max(1, len(durs.get((k, g), [])))
        rows.append((k[:34], g, m))
    out[MAP] = rows
    print("== SMN_XCD_MAP=%d" % MAP)
    for k, g, m in rows:
        hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
        print("%-34s grid %5d  %8.1f us  FETCH %7.1f MB  WRITE %7.1f MB  L2 hit %.3f  EA_RDREQ %.3g (32B %.3g)  TCC_REQ %.3g READ %.3g" % (
            k, g, m["us"], m.get("FETCH_SIZE", 0) / 1024, m.get("WRITE_SIZE", 0) / 1024, hit, m.get("TCC_EA0_RDREQ_sum", 0),
            m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_REQ_sum", 0), m.get("TCC_READ_sum", 0)))
json.dump({str(k): [(a, b, c) for a, b, c in v] for k, v in out.items()}, open("$O/l2_summary.json", "w"), indent=1)
PY
