#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
O=$R/gpurun_out/r04human30; mkdir -p $O
export TMPDIR=/tmp
( time WTZ_BENCH_BACKEND=gloo timeout 1100 python bench.py --gpus 2 --workload human30 --steps 1 --warmup 0 --no-cpu-baseline --pool-gb 48 ) > $O/bench_human30_2ranks.json 2> $O/bench_human30_2ranks.err
tail -n 1 $O/bench_human30_2ranks.json | cut -c1-900; grep "real\|records,\|host seconds\|kernel ms" $O/bench_human30_2ranks.err | tail -n 5
