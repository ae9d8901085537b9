#!/bin/bash
R=$GRAFT_REPO_ROOT
export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_f1half.so
run() { env "$@" python3 $R/bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $*  %.3f ms/step  chol %.3f panel %.3f trail %.3f' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['trail']))"; }
for round in 1 2; do
  CFG=c4; run SMN_NONE=1; for t in 1000 1700 2700 4000; do run SMN_EXPERIMENT_F1HALF=$t; done
done
