#!/bin/bash
# round 5: configs[3] shape with the all-reads z-mer index beside a pool sized from the input
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05u}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|host seconds\|kernel ms\|wall seconds\|real\|z-mer index\|batches in\|splitting\|failed\|error" $O/bench_fly70.err | tail -14
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
