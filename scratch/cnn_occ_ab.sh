#!/bin/bash
# conv-NNGP pair kernel: workgroups per CU the 32x32 forms are compiled for (variant builds made by the caller)
for v in "" _c3 _c2; do
  lib=scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp$v.so
  [ -f $lib ] || continue
  echo "== $lib"
  SMNNGP_LIB=$PWD/$lib timeout -k 10 300 python scratch/cnn_probe.py
done
