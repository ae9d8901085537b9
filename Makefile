# Convenience targets; everything is plain Python + hipcc underneath (see README.md).
PY ?= python

.PHONY: build test test-gpu bench golden clean

build:            ## hipcc --offload-arch=gfx950 -> in-tree libsmnngp.so (cross-compiles without a GPU)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU: oracle, ABI, host logic, gloo world_size-2
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## MI355X: parity against the oracle through the C ABI
	$(PY) -m pytest tests -q -m gpu

bench:            ## one JSON line: N=16384 d=3072 4-layer ReLU NNGP + Cholesky + LML, fp32
	$(PY) bench.py

golden:           ## regenerate tests/golden/nngp_golden.npz from the oracle
	$(PY) tests/golden/make_golden.py

clean:
	rm -rf scale-mixtures-of-neural-network-gaussian-processes_amd/build* scale-mixtures-of-neural-network-gaussian-processes_amd/*.so scratch/bf16x3/*.so scratch/valu_rate/valu_rate scratch/dpp_probe/dpp_probe
