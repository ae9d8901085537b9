#!/bin/bash
# A/B: strips of K >= 256 (512-column outer panels) through the pipelined K loop (variant libsmnngp_striptag.so) or the plain one.
R=$GRAFT_REPO_ROOT
P=$R/scale-mixtures-of-neural-network-gaussian-processes_amd
for round in 1 2 3; do
  for lib in libsmnngp.so libsmnngp_striptag.so; do
    for cfg in c4 c5; do
      SMNNGP_LIB=$P/$lib python3 $R/bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $round $lib $cfg  %.3f ms/step  chol %.3f  strip %.3f trail %.3f' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['strip'], d['phases_ms']['trail']))"
    done
  done
done
