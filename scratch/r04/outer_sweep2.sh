#!/bin/bash
R=$GRAFT_REPO_ROOT
export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_outer.so
run() { env "$@" python3 $R/bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $*  %.3f ms/step  chol %.3f  panel %.3f strip %.3f trail %.3f' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['strip'], d['phases_ms']['trail']))"; }
for round in 1 2; do
  CFG=c4
  run SMN_EXPERIMENT_OUTER=256
  run SMN_EXPERIMENT_OUTER=512
  run SMN_EXPERIMENT_OUTER=1024
  run SMN_EXPERIMENT_OUTER=512 SMN_SUPER=2048
  run SMN_EXPERIMENT_OUTER=512 SMN_SUPER=1536
  run SMN_EXPERIMENT_OUTER=768 SMN_SUPER=1536
  CFG=c5
  run SMN_EXPERIMENT_OUTER=256
  run SMN_EXPERIMENT_OUTER=512
done
CFG=c4
python3 $R/bench.py --dtype f64 --n 8192 --steps 6 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 n8192 outer=256 %.3f' % d['ms_per_step'])"
SMN_EXPERIMENT_OUTER=512 python3 $R/bench.py --dtype f64 --n 8192 --steps 6 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 n8192 outer=512 %.3f' % d['ms_per_step'])"
