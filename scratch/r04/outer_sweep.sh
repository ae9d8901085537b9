#!/bin/bash
# EXPERIMENT: outer-panel width of the two-level Cholesky (variant build libsmnngp_outer.so reads SMN_EXPERIMENT_OUTER), C4 and C2.
R=$GRAFT_REPO_ROOT
export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp_outer.so
for round in 1 2; do
  for w in 256 128 384 512; do
    for cfg in c4 c2; do
      SMN_EXPERIMENT_OUTER=$w python3 $R/bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $round outer=$w $cfg  %.3f ms/step  chol %.3f  panel %.3f strip %.3f trail %.3f  logpdf %r' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['strip'], d['phases_ms']['trail'], d['result']['logpdf']))"
    done
  done
done
