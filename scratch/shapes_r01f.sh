#!/bin/bash
# bench lines at the other BASELINE shapes (C2, C5 shape, fp64) with the r01f build
B="--steps 10 --warmup 3 --no-cpu-baseline"
timeout -k 10 300 python bench.py $B --n 4096 --d 512 --layers 3 > gpurun_out/r01f_c2.json 2> gpurun_out/r01f_c2.err
timeout -k 10 300 python bench.py $B --n 32768 --d 1024 --layers 6 --act erf > gpurun_out/r01f_c5.json 2> gpurun_out/r01f_c5.err
timeout -k 10 300 python bench.py $B --dtype f64 --n 8192 > gpurun_out/r01f_f64.json 2> gpurun_out/r01f_f64.err
timeout -k 10 300 python scratch/rec_probe.py > gpurun_out/r01f_rec_probe.txt 2>&1
for f in c2 c5 f64; do python - <<PY
import json
d=json.load(open("gpurun_out/r01f_$f.json"))
print("$f", round(d["ms_per_step"],3), d["phases_ms"], {k[:20]:round(v["frac"],3) for k,v in d["roofline_other_kernels"].items() if "frac" in v}, round(d["roofline"]["frac"],3), d["roofline"].get("frac_exclusive"))
PY
done
tail -30 gpurun_out/r01f_rec_probe.txt
