#!/bin/bash
# potrf phase timing + a sweep of the schedule knobs (block width S, window D, reserved CUs R) at C4
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_pt.so timeout -k 5 120 python3 scratch/panel_timing.py 2>&1 | grep -E "potrf|info" | tee $O/potrf_phases.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t1.log 2>&1
echo "pytest rc=$?"; tail -3 $O/t1.log
run() {
  echo -n "$* : "
  env "$@" timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.3f ms/step  chol %.3f  panel %.2f trail %.2f logpdf %.4f'%(d['ms_per_step'],d['roofline']['cholesky_wall_ms'],d['phases_ms']['panel'],d['phases_ms']['trail'],d['result']['logpdf']))"
}
for S in 1024 512; do for D in 2 3 5 1000; do for C in 32 16; do
  run SMN_SUPER=$S SMN_WINDOW=$D SMN_CHAIN_CUS=$C
done; done; done 2>&1 | tee $O/sweep.txt
