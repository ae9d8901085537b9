#!/bin/bash
# r03 experiment (not shipped): applies the early-pass variant of panelr_kernel to cholesky.hip; A/B in profiles/r03_panel_early_pass_ab.txt
# (build the variants with build.build(variant="emN", defines=("-DSMN_PANEL_EARLY_MFMA=N",)) first)
# r03: helper waves on a free SIMD pre-apply the K < cb part of the next block's update beside the leaf: budget sweep
P=scale-mixtures-of-neural-network-gaussian-processes_amd
mkdir -p gpurun_out
python3 scratch/r03/dbg_chol.py > gpurun_out/early_dbg.txt 2>&1 || exit 1
for v in "" _em0 _em24 _em56 _em80; do
  echo "== variant '${v}'" 
  SMNNGP_LIB=$PWD/$P/libsmnngp${v}.so python3 bench.py --config c2 --steps 200 --warmup 20 --no-other-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c2', d['ms_per_step'])"
  SMNNGP_LIB=$PWD/$P/libsmnngp${v}.so python3 bench.py --steps 20 --warmup 3 --no-other-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c4', d['ms_per_step'], d['roofline'].get('cholesky_wall_ms'))"
  SMNNGP_LIB=$PWD/$P/libsmnngp${v}.so python3 scratch/r03/f64_chol.py
done
SMNNGP_LIB=$PWD/$P/libsmnngp_timing.so python3 scratch/r03/panel_timing.py 2048 float32
