#!/bin/bash
TAG=${1:-r03q}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_dp_forms.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline --steps 3 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-200; grep "kernel ms" $O/bench.err | tail -1
python bench.py --workload ecoli --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_ecoli.json 2> $O/bench_ecoli.err; tail -1 $O/bench_ecoli.json | cut -c1-200; grep "kernel ms" $O/bench_ecoli.err | tail -1
