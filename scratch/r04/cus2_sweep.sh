#!/bin/bash
# EXPERIMENT: a second, wider CU reservation for the F1 launches of the chain-bound phase (variant libsmnngp_cus2.so).
R=$GRAFT_REPO_ROOT
export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_cus2.so
run() { env "$@" python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*  %.3f ms/step  chol %.3f panel %.3f trail %.3f  %r' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['trail'], d['result']['logpdf']))"; }
for round in 1 2; do
  run SMN_NONE=1
  for c in 48 64 96; do for t in 1700 2700 4000 5500; do run SMN_EXPERIMENT_CUS2=$c SMN_EXPERIMENT_TILES2=$t; done; done
done
