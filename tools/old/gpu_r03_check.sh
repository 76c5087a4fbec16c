#!/bin/bash
# Re-entry check on the GPU box: the GPU test-suite, the driver's exact bench command, rocprofv3 kernel stats of a zmo step.
# usage: tools/gpu_r03_check.sh <tag>
TAG=${1:-r03c}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | head -2; grep real $O/pytest_gpu.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_zmo.json 2> $O/bench_zmo.err
tail -1 $O/bench_zmo.json | cut -c1-300; grep real $O/bench_zmo.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/trace_zmo.log 2>&1
cd $R
python tools/summarize_profiles.py $O $O/summary
find $O -name "*kernel_trace.csv" -size +8M -delete
ls -la $O/summary
