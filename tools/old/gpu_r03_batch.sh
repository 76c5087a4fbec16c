#!/bin/bash
TAG=${1:-r03batch}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
for mb in 2048 4096 8192 16384; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 2 --max-batch $mb > $O/bench_$mb.json 2> $O/bench_$mb.err
  echo "max-batch $mb: $(tail -1 $O/bench_$mb.json | cut -c88-190)"; grep -E "host seconds|batches in" $O/bench_$mb.err | tail -2 | cut -c1-230
done
