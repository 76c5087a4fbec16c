#!/bin/bash
# quick look: E. coli-shape bench with the stage profile of the lane pipelines
TAG=${1:-r03c}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
WTZ_PROFILE_PAIR=1 python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 $2 > $O/bench_ecoli.json 2> $O/bench_ecoli.err; tail -1 $O/bench_ecoli.json | cut -c1-200; grep -E "lane-profile" $O/bench_ecoli.err | tail -4; grep "kernel ms" $O/bench_ecoli.err | tail -1; grep -E "align-profile" $O/bench_ecoli.err | tail -2
md5sum /tmp/wtz_bench/bench_r0.ovl 2>/dev/null
