#!/bin/bash
# EXPERIMENT: the bulk far update F1 in 64x128 half tiles (24 KB of LDS, 152 VGPRs: one such workgroup can share a CU with a panel workgroup).
R=$GRAFT_REPO_ROOT
export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_f1half.so
run() { env "$@" python3 $R/bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $*  %.3f ms/step  chol %.3f panel %.3f trail %.3f  %r' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['trail'], d['result']['logpdf']))"; }
for round in 1 2; do
  CFG=c4; run SMN_NONE=1; run SMN_EXPERIMENT_F1HALF=1; run SMN_EXPERIMENT_F1HALF=1 SMN_CHAIN_CUS=16; run SMN_EXPERIMENT_F1HALF=1 SMN_CHAIN_CUS=0
  CFG=c5; run SMN_NONE=1; run SMN_EXPERIMENT_F1HALF=1
done
