P=$PWD/scale-mixtures-of-neural-network-gaussian-processes_amd
for lib in libsmnngp.so libsmnngp_s1.so; do for map in 0 1; do for lds in 0 100000; do
SMNNGP_LIB=$P/$lib SMN_XCD_MAP=$map SMN_DEBUG_LDS=$lds python scratch/gemm_probe.py 2>&1 | tail -1
done; done; done
SMN_XCD_MAP=0 SMN_DEBUG_LDS=60000 SMNNGP_LIB=$P/libsmnngp_s1.so python scratch/gemm_probe.py | tail -1
SMN_XCD_MAP=0 SMN_DEBUG_LDS=45000 SMNNGP_LIB=$P/libsmnngp_s1.so python scratch/gemm_probe.py | tail -1
