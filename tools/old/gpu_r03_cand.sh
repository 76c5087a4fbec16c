#!/bin/bash
TAG=${1:-r03l}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_ecoli.json 2> $O/bench_ecoli.err; grep "kernel ms" $O/bench_ecoli.err | tail -1; grep records $O/bench_ecoli.err | tail -1
python bench.py --workload ecoli --engine dmo --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_ecoli_dmo.json 2> $O/bench_ecoli_dmo.err; grep "kernel ms" $O/bench_ecoli_dmo.err | tail -1; grep records $O/bench_ecoli_dmo.err | tail -1
python bench.py --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_yeast.json 2> $O/bench_yeast.err; tail -1 $O/bench_yeast.json | cut -c1-180; grep "kernel ms" $O/bench_yeast.err | tail -1; grep records $O/bench_yeast.err | tail -1
md5sum /tmp/wtz_bench/*.ovl 2>/dev/null
