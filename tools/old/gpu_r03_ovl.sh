#!/bin/bash
TAG=${1:-r03n}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
for cfg in WTZ_RANGE_OVERLAP=1 WTZ_RANGE_OVERLAP=0; do
  env $cfg python bench.py --no-cpu-baseline --steps 3 --warmup 3 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  echo $cfg; tail -1 $O/bench_$cfg.json | cut -c1-180; grep "kernel ms" $O/bench_$cfg.err | tail -1; grep -E "records|host seconds|batches in" $O/bench_$cfg.err | tail -3
done
python bench.py --engine dmo --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_dmo.json 2> $O/bench_dmo.err; tail -1 $O/bench_dmo.json | cut -c1-180; grep -E "records|host seconds|batches in" $O/bench_dmo.err | tail -3
