#!/bin/bash
# A/B of build-time / env variants of the GEMM kernels on the full bench workload
P="scale-mixtures-of-neural-network-gaussian-processes_amd"
for lib in libsmnngp.so libsmnngp_s1.so; do
  for map in 1 0; do
    SMNNGP_LIB=$PWD/$P/$lib SMN_XCD_MAP=$map timeout -k 10 300 python bench.py --no-cpu-baseline --no-recursion-probe --steps 5 --warmup 2 > gpurun_out/ab_${lib}_${map}.json 2> gpurun_out/ab_${lib}_${map}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_${lib}_${map}.json"))
print("${lib} map=${map}", round(d["ms_per_step"],3), d["phases_ms"], round(d["roofline"]["frac"],3), d["result"]["logdet"])
PY
  done
done
