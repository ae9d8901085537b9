#!/bin/bash
# The f32 tile engine on v_mfma_f32_16x16x4_f32 (variant build m16) against the default 32x32x2 form: parity suite on the
# variant, then interleaved bench rounds (C4), C5 and C2 once each.
set -e
P=scale-mixtures-of-neural-network-gaussian-processes_amd
SMNNGP_LIB=$PWD/$P/libsmnngp_m16.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -4
one() {
  local lib=$PWD/$P/libsmnngp$1.so; shift
  SMNNGP_LIB=$lib timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; o=j['roofline_other_kernels']
print('   ms/step %.3f  chol %.3f  build %.3f  frac_excl %s  logpdf %.6f' % (j['ms_per_step'], r.get('cholesky_wall_ms',0), j['phases_ms']['build'], r.get('frac_exclusive'), j['result']['logpdf']))"
}
for round in 1 2 3; do
  for v in "" _m16; do echo "round $round variant '${v}' C4"; one "$v" --steps 20 --warmup 3; done
done
for v in "" _m16; do echo "variant '${v}' C5"; one "$v" --config c5 --steps 5 --warmup 1; done
for v in "" _m16; do echo "variant '${v}' C2"; one "$v" --config c2 --steps 50 --warmup 5; done
