#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t1.log 2>&1
echo "pytest rc=$?"; tail -3 $O/t1.log
run() {
  echo -n "$* : "
  env "$@" timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.3f ms/step  chol %.3f  panel %.2f trail %.2f logpdf %.4f'%(d['ms_per_step'],d['roofline']['cholesky_wall_ms'],d['phases_ms']['panel'],d['phases_ms']['trail'],d['result']['logpdf']))"
}
(
for F in 1000000000 4096 8192; do for D in 1 2 3 1000; do for C in 32 16; do
  run SMN_FUSED_ROWS=$F SMN_WINDOW=$D SMN_CHAIN_CUS=$C
done; done; done
run SMN_FUSED_ROWS=1000000000 SMN_WINDOW=2 SMN_SUPER=512
run SMN_FUSED_ROWS=1000000000 SMN_WINDOW=4 SMN_SUPER=512
run SMN_FUSED_ROWS=4096 SMN_WINDOW=2 SMN_CHAIN_CUS=8
) 2>&1 | tee $O/sweep.txt
