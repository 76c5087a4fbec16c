#!/bin/bash
# configs[2] bench (2 timed steps) with the lane pipelines' stage profile; $2 = extra env assignments, e.g. "WTZ_GAP_LANE=0"
TAG=${1:-r03y}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
for cfg in "" $2; do
  n=${cfg:-default}
  env $cfg WTZ_PROFILE_PAIR=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err
  tail -1 $O/bench_$n.json | cut -c1-180; grep "kernel ms" $O/bench_$n.err | tail -1; grep -E "records" $O/bench_$n.err | tail -1
done
grep -E "lane-profile\] [0-9]+ windows" $O/bench_default.err | sed 's/K-sw1 problems.*chained kernel;//' | tail -4
grep -E "lane-profile\] [0-9]+ window slots" $O/bench_default.err | tail -3
