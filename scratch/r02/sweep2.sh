#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
run() {
  echo -n "$* : "
  env "$@" timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.3f ms/step  chol %.3f  panel %.2f trail %.2f logpdf %.4f'%(d['ms_per_step'],d['roofline']['cholesky_wall_ms'],d['phases_ms']['panel'],d['phases_ms']['trail'],d['result']['logpdf']))"
}
(
run SMN_CHAIN_CUS=0 SMN_WINDOW=1000
run SMN_CHAIN_CUS=0 SMN_WINDOW=1000 SMN_SUPER=512
run SMN_CHAIN_CUS=0 SMN_WINDOW=3 SMN_MAXK=1024
run SMN_CHAIN_CUS=0 SMN_WINDOW=3 SMN_MAXK=2048
run SMN_CHAIN_CUS=0 SMN_WINDOW=1 SMN_MAXK=1024
run SMN_CHAIN_CUS=0 SMN_WINDOW=3
run SMN_CHAIN_CUS=0 SMN_WINDOW=1000 SMN_PERSISTENT=0
run SMN_CHAIN_CUS=8 SMN_WINDOW=1000
run SMN_CHAIN_MIN_N=100000000
) 2>&1 | tee $O/sweep.txt
