#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
bash scratch/r02/samepanel_ab.sh 2>&1 | tee $O/samepanel_ab.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pipelined or rccl or shard or lower_blocks" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -5 $O/t.log
timeout -k 10 300 python3 bench.py --sharded-path --no-cpu-baseline > $O/bench_sharded_1rank.json 2> $O/bench_sharded.err
echo "sharded bench rc=$?"; tail -3 $O/bench_sharded.err
python3 -c "
import json;d=json.load(open('$O/bench_sharded_1rank.json'));print(d['ms_per_step'],d['phases_ms'],{k:d[k] for k in d if 'exchange' in k or 'build' in k}, d['result'])"
timeout -k 10 300 python3 bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
echo "c2 rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_c2.json'));print(d['ms_per_step'],d['phases_ms'],d['executed_tflops'])"
