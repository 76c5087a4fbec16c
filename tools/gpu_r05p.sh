#!/bin/bash
# round 5: dmo band list as a register-chained sequence (64 starts searched at once): dmo golden parity, both engines with md5, phase profile
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05p}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dmo" > $O/pytest_parity_dmo.txt 2>&1; tail -2 $O/pytest_parity_dmo.txt
for e in dmo zmo; do
  timeout 600 python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
  grep "kernel ms" $O/bench_$e.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1])
print('$e', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
done
bash tools/gpu_phase_profile.sh $T
