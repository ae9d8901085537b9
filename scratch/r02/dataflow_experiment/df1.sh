#!/bin/bash
# first runs of the dataflow Cholesky: small factorisations against LAPACK, then the parity tests, then the bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02_df1}
mkdir -p $O
cd $R
export SMN_DATAFLOW=1
timeout -k 10 120 python3 - <<'PY' > $O/small.txt 2>&1
import ctypes as C, numpy as np, sys, time
sys.path.insert(0, '.')
from smnngp import _lib as L
ctx = L.Context(0)
for n, m in ((256, 128), (512, 0), (1024, 128), (2048, 128), (4096, 128)):
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n + m, 64)); a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(np.float32)
    ad = ctx.to_device(a)
    info, logdet = C.c_int(), C.c_double()
    t0 = time.time()
    ctx.call("smn_cholesky", L.F32, ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    dt = time.time() - t0
    l = np.linalg.cholesky(a[:n, :n].astype(np.float64))
    got = ad.numpy().astype(np.float64)
    e1 = np.abs(np.tril(got[:n, :n]) - l).max() / np.abs(l).max()
    print("n=%d m=%d info=%d logdet err %.2e  L err %.2e  (%.1f ms first call)" % (n, m, info.value, abs(logdet.value - 2 * np.log(np.diag(l)).sum()) / abs(logdet.value), e1, dt * 1e3), flush=True)
PY
echo "small rc=$?"; cat $O/small.txt
grep -q "info=0" $O/small.txt || exit 1
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -6 $O/t.log
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['phases_ms'],d['roofline']['cholesky_wall_ms'], d['result'])"
