#!/bin/bash
# Do the L2 misses of the trailing update cost time?  A/B: the real kernel against a debug build in which every tile reads
# one of 8 operand panels (everything hits L2; results wrong, timing only).  Look-ahead off: nothing else on the GPU.
set -u
R=$GRAFT_REPO_ROOT
cd $R
export SMN_CHAIN_MIN_N=1000000000
for lib in libsmnngp.so libsmnngp_samepanel.so; do
  for i in 1 2; do
  echo -n "$lib : "
  SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/$lib timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.3f ms/step  trail(summed, exclusive) %.3f ms  frac %.3f  build %.3f'%(d['ms_per_step'],d['phases_ms']['trail'],d['roofline']['frac'],d['phases_ms']['build']))"
  done
done
