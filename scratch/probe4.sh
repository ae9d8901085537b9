P=$PWD/scale-mixtures-of-neural-network-gaussian-processes_amd
for lib in libsmnngp.so libsmnngp_noslp.so; do for l in 0 4 8; do echo -n "$lib "; SMNNGP_LIB=$P/$lib PL=$l python scratch/rec_probe.py | tail -1; done; done
SMNNGP_LIB=$P/libsmnngp_noslp.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-recursion-probe > gpurun_out/b_noslp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/b_noslp.json'));print('noslp bench', d['ms_per_step'], d['phases_ms'])"
