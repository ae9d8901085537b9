#!/bin/bash
# conv-NNGP on 32x32 images: register-only stencil (conv_pair32_kernel) against the LDS-map kernel
for t in 1 0 1 0; do echo "== SMN_CNN_FAST32=$t"; SMN_CNN_FAST32=$t timeout -k 10 300 python scratch/cnn_probe.py; done
