#!/bin/bash
P="$PWD/scale-mixtures-of-neural-network-gaussian-processes_amd"
run() { # lib lookahead
  SMNNGP_LIB=$P/$1 SMN_LOOKAHEAD=$2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-recursion-probe --steps 5 --warmup 2 > gpurun_out/ab2_$1_$2.json 2> gpurun_out/ab2_$1_$2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab2_$1_$2.json"))
print("$1 la=$2", round(d["ms_per_step"],3), d["phases_ms"], d["result"]["logdet"], d["result"]["info"])
PY
}
run libsmnngp.so 0
run libsmnngp_xr32.so 0
run libsmnngp_la.so 0
run libsmnngp_la.so 1
run libsmnngp_s1.so 1
run libsmnngp.so 1
