#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02_df2}
mkdir -p $O
cd $R
export SMN_DATAFLOW=1
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -6 $O/t.log
for cfg in c4 c2; do
SMN_DF_DEBUG=1 timeout -k 10 200 python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/bench_$cfg.json 2> $O/bench_$cfg.err
echo "bench $cfg rc=$?"; grep "dataflow plan" $O/bench_$cfg.err | head -2; python3 -c "
import json;d=json.load(open('$O/bench_$cfg.json'));print(d['ms_per_step'],d['phases_ms'],d['roofline']['cholesky_wall_ms'], d['result'])"
done
