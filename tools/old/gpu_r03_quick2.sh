#!/bin/bash
TAG=${1:-r03g}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_dp_forms.py -x -q -m gpu -k "fixed or global" > $O/pytest_dp.log 2>&1; tail -3 $O/pytest_dp.log
WTZ_PROFILE_PAIR=1 python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_ecoli.json 2> $O/bench_ecoli.err; tail -1 $O/bench_ecoli.json | cut -c1-200; grep -E "lane-profile" $O/bench_ecoli.err | sed 's/K-sw1 problems.*chained kernel;//' | tail -6; grep "kernel ms" $O/bench_ecoli.err | tail -1
WTZ_WINALIGN_LANE=2 python bench.py --workload ecoli --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_ecoli_check.json 2> $O/bench_ecoli_check.err; tail -1 $O/bench_ecoli_check.json | cut -c1-120; grep -E "differs|failed" $O/bench_ecoli_check.err | head -3
