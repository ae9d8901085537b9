#!/bin/bash
# conv-NNGP pair kernel: XCD-tiled pair order on / off
for t in 1 0 1 0; do echo "== SMN_CNN_TILED=$t"; SMN_CNN_TILED=$t timeout -k 10 300 python scratch/cnn_probe.py; done
