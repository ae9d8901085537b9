#!/bin/bash
# 16-row f32 panel workgroups for the last rows against 64-row ones.  The variants were builds of a working tree that took the threshold
# as a macro (libsmnngp_tNNNN.so = 16-row workgroups from NNNN rows down); what was kept: kPanelSmallRows = 4096, kPanelSmallXR = 16.
R=$PWD
O=$R/gpurun_out/r04g
mkdir -p $O
F="--no-cpu-baseline --no-recursion-probe --no-exclusive-probe --no-other-workloads"
for round in 1 2; do
for v in "" _t4096 _t6144 _t8192; do
  export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp$v.so
  c2=$(timeout -k 10 100 python3 bench.py --config c2 --steps 300 --warmup 20 $F | python3 -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  c4=$(timeout -k 10 100 python3 bench.py --steps 20 --warmup 3 $F | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  n8=$(timeout -k 10 100 python3 bench.py --n 8192 --steps 40 --warmup 5 $F | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  echo "round $round variant '${v}': C2 $c2 ms  C4 $c4 ms  N=8192 $n8 ms"
done
done
for v in; do
  export SMNNGP_LIB=$R/scale-mixtures-of-neural-network-gaussian-processes_amd/libsmnngp$v.so
  echo "small_n_latency variant '${v}'"
  timeout -k 10 100 python3 scratch/small_n_latency.py | grep -E "loss  |loss_and_grad" 
done
