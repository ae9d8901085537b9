#!/bin/bash
for env in "SMN_PERSISTENT=0" "SMN_XCD_MAP=0" "SMN_CHAIN_CUS=0" "SMN_PANEL_LEAF=0" "SMN_SUPER=256" "SMN_CHAIN_MIN_N=1024"; do
  echo "== $env"
  env $env python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grad or predict or chol or lml or live_tiles" 2>&1 | tail -2
done
