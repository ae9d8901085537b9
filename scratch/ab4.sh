#!/bin/bash
run() { # super
  SMN_SUPER=$1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-recursion-probe --steps 5 --warmup 2 > gpurun_out/ab4_$1.json 2> gpurun_out/ab4_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab4_$1.json"))
print("super=$1", round(d["ms_per_step"],3), d["phases_ms"], round(d["roofline"]["frac"],3), d["result"]["logdet"], d["result"]["info"])
PY
}
run 0
run 1024
run 2048
run 4096
run 8192
