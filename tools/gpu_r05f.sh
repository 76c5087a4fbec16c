#!/bin/bash
# round 5: trace byte without the "bases equal" bit (mat / mis from the score): DP vectors, 40 000 jobs against the round-4 kernel, whole step with md5; phase profile of the pair kernels after the pool change
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05f}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_dp_forms.py -x -q -k "shift" > $O/pytest_shift.txt 2>&1; tail -3 $O/pytest_shift.txt
timeout 300 python tools/ubench/ksw3_bench.py --forms 1,5,0 --reps 2 > $O/ksw3_bench.txt 2> $O/ksw3_bench.err; cat $O/ksw3_bench.txt
for e in zmo dmo; do
  timeout 600 python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
  grep "kernel ms" $O/bench_$e.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1])
print('$e', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('parity'))
"
done
timeout 400 python bench.py --workload ecoli --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; python3 -c "
import json
d=json.loads(open('$O/bench_ecoli.json').read().strip().split('\n')[-1])
print('ecoli', d['ms_per_step'], d['value'], d.get('parity'))
"
bash tools/gpu_phase_profile.sh $T
