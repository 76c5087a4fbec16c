#!/bin/bash
# round 5, re-entry baseline: K-sw3 in isolation (forms 1, 5, product), both engines at configs[2] with md5, configs[1], then the profile refresh (kernel stats + PMC) of both engines
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05g}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python tools/ubench/ksw3_bench.py --forms 1,5,0 --reps 2 > $O/ksw3_bench.txt 2> $O/ksw3_bench.err; cat $O/ksw3_bench.txt
for e in zmo dmo; do
  timeout 600 python bench.py --engine $e --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
  grep "kernel ms" $O/bench_$e.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1])
print('$e', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('parity'))
print({a:round(b) for a,b in d.get('kernel_ms_last_step',{}).items()})
"
done
timeout 400 python bench.py --workload ecoli --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; python3 -c "
import json
d=json.loads(open('$O/bench_ecoli.json').read().strip().split('\n')[-1])
print('ecoli', d['ms_per_step'], d['value'], d.get('parity'))
"
bash tools/gpu_r05_refresh.sh $T zmo
bash tools/gpu_r05_refresh.sh $T dmo
