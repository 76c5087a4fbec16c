#!/bin/bash
# round-2 GPU check: the whole -m gpu suite (no -x: see every failure), then a short E. coli-shape bench line for regressions
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.log 2>&1
tail -40 $O/pytest_gpu.log
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_ecoli.json 2> $O/bench_ecoli.err
tail -1 $O/bench_ecoli.json | cut -c1-900
grep -E "kernel ms" $O/bench_ecoli.err | tail -1
