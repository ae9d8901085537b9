#!/bin/bash
# final conv-NNGP numbers of the r01f build: default, fast32 forced for fp32 too, LDS-map kernel only, plain pair order
for e in "SMN_NOP=1" "SMN_CNN_FAST32=2" "SMN_CNN_FAST32=0" "SMN_CNN_TILED=0"; do echo "== $e"; env $e timeout -k 10 300 python scratch/cnn_probe.py; done
