#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py --genome 460000 --steps 1 --warmup 0 --cpu-genome 230000 ) > gpurun_out/bench_small.log 2>&1
tail -3 gpurun_out/bench_small.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -o small -- python $GRAFT_REPO_ROOT/bench.py --genome 460000 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_small.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_small | head -20
( time timeout 2400 python bench.py --steps 1 --warmup 0 ) > gpurun_out/bench_ecoli.log 2>&1
tail -3 gpurun_out/bench_ecoli.log
