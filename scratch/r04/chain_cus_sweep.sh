#!/bin/bash
# C4 step time against the number of CUs the bulk far update may NOT use (SMN_CHAIN_CUS), two rounds.
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for c in 32 16 24 40 48 64 96; do
    SMN_CHAIN_CUS=$c python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $round SMN_CHAIN_CUS=$c  %.3f ms/step  chol %.3f  panel %.3f  trail %.3f' % (d['ms_per_step'], d['roofline']['cholesky_wall_ms'], d['phases_ms']['panel'], d['phases_ms']['trail']))"
  done
done
