#!/bin/bash
# A/B of the tile engine's global-load placement (variant builds): default | loads right after the staging writes (ge) |
# staging writes one group later (wg2).  Interleaved rounds in one call; C4 step, then the C5 shape once each.
set -e
P=scale-mixtures-of-neural-network-gaussian-processes_amd
one() {  # $1 = lib suffix ("" = default), rest = bench args
  local lib=$P/libsmnngp$1.so; shift
  SMNNGP_LIB=$lib timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-recursion-probe 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   ms/step %.3f  chol %.3f  build %.3f  frac_excl %s  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], r.get('frac_exclusive'), j['result']['logpdf']))"
}
for round in 1 2 3; do
  for v in "" _ge _wg2; do echo "round $round variant '${v}' C4"; one "$v" --steps 20 --warmup 3; done
done
for v in "" _ge _wg2; do echo "variant '${v}' C5"; one "$v" --config c5 --steps 5 --warmup 1; done
