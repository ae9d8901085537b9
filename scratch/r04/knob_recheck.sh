#!/bin/bash
# With 512-column outer panels: do the remaining schedule switches still sit at their optimum?  C4, ms per step.
R=$GRAFT_REPO_ROOT
run() { env "$@" python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*  %.3f ms/step  chol %.3f' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms']))"; }
for round in 1 2; do
  run SMN_NONE=1
  run SMN_CHAIN_CUS=24
  run SMN_CHAIN_CUS=40
  run SMN_SUPER=1536
  run SMN_SUPER=2048
  run SMN_SUPER_WIDE_ROWS=12288
  run SMN_SUPER_WIDE_ROWS=8192
  run SMN_XCD_MAP=0
done
