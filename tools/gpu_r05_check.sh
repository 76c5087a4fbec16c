#!/bin/bash
# round 5: a short check of the last host-only edit (golden parity of both engines, configs[2] with md5, configs[1])
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05check}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden" > $O/pytest_parity.txt 2>&1; tail -1 $O/pytest_parity.txt
for w in "" "--workload ecoli" "--engine dmo"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $w > $O/b.json 2> $O/b.err
  python3 -c "
import json
d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$w', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
done
