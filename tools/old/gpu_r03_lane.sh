#!/bin/bash
# lane-per-problem K-sw1 (wtz_sw_lane.h): function-level vectors, goldens with the on-device cross-check (WTZ_WINALIGN_LANE=2: lane pipeline vs
# chained kernel on every window), then the bench lines with the lane form on / off
TAG=${1:-r03b}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_dp_forms.py -x -q -m gpu -k fixed > $O/pytest_dp.log 2>&1; tail -3 $O/pytest_dp.log
WTZ_WINALIGN_LANE=2 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity_check.log 2>&1; tail -3 $O/pytest_parity_check.log
WTZ_PROFILE_PAIR=1 python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_ecoli_lane.json 2> $O/bench_ecoli_lane.err; tail -1 $O/bench_ecoli_lane.json | cut -c1-200; grep -E "lane-profile" $O/bench_ecoli_lane.err | tail -3; grep "kernel ms" $O/bench_ecoli_lane.err | tail -1
WTZ_WINALIGN_LANE=2 python bench.py --workload ecoli --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_ecoli_check.json 2> $O/bench_ecoli_check.err; tail -1 $O/bench_ecoli_check.json | cut -c1-120; grep -E "differs|failed" $O/bench_ecoli_check.err | head -3
WTZ_WINALIGN_LANE=0 python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_ecoli_old.json 2> $O/bench_ecoli_old.err; grep "kernel ms" $O/bench_ecoli_old.err | tail -1
python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_yeast_lane.json 2> $O/bench_yeast_lane.err; tail -1 $O/bench_yeast_lane.json | cut -c1-200; grep "kernel ms" $O/bench_yeast_lane.err | tail -1; grep -E "records" $O/bench_yeast_lane.err | tail -1
