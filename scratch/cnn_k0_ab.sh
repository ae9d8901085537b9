#!/bin/bash
for v in "" _rr "" _rr; do
  lib=scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp$v.so
  echo "== $lib"
  SMNNGP_LIB=$PWD/$lib timeout -k 10 300 python scratch/cnn_probe.py
done
