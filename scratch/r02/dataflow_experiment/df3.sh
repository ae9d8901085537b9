#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02_df3}
mkdir -p $O
cd $R
export SMN_DATAFLOW=1
timeout -k 5 60 python3 scratch/r02/df_small.py 2>&1 | grep -v Warn | grep "n=" || exit 1
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or trsm or lml or predict or spr" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -4 $O/t.log
SMN_DF_TRACE=$O/trace_c4.bin timeout -k 5 100 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > /dev/null 2>&1; python3 scratch/r02/df_trace.py $O/trace_c4.bin
for cfg in c4 c2; do
SMN_DF_DEBUG=1 timeout -k 10 200 python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recursion-probe --no-exclusive-probe > $O/bench_$cfg.json 2> $O/bench_$cfg.err
echo "bench $cfg rc=$?"; grep "dataflow plan" $O/bench_$cfg.err | head -1; python3 -c "
import json;d=json.load(open('$O/bench_$cfg.json'));print(d['ms_per_step'],d['phases_ms']['build'],d['roofline']['cholesky_wall_ms'], d['result'])"
done
