#!/bin/bash
# round 5: the sharded bump pool (wtz_pool_alloc, slabs per workgroup shard) + the frame form of K-sw3: GPU suite, K-sw3 in isolation, both engines at configs[2] and configs[1]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05e}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python tools/ubench/ksw3_bench.py --forms 1,5,2,0 --reps 2 > $O/ksw3_bench.txt 2> $O/ksw3_bench.err; cat $O/ksw3_bench.txt
for e in zmo dmo; do
  timeout 600 python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
  grep "kernel ms" $O/bench_$e.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1])
print('$e', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('parity'))
"
done
WTZ_SW_MW_MIN=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_zmo_nomw.json 2> $O/bench_zmo_nomw.err; grep "kernel ms" $O/bench_zmo_nomw.err | tail -1
WTZ_SW_MW_MIN=2048 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_zmo_mw2048.json 2> $O/bench_zmo_mw2048.err; grep "kernel ms" $O/bench_zmo_mw2048.err | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
