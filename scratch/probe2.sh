for d in 256 512 1024 3072; do PD=$d python scratch/gemm_probe.py | tail -1; done
