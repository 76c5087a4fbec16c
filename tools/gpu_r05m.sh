#!/bin/bash
# round 5: the configs[3] shape (9.8 Gbp of reads, per-batch z-index) as a bench line with its reference stripe, after the seed-lookup work
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05m}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|host seconds\|kernel ms\|wall seconds\|real" $O/bench_fly70.err | tail -8
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity'))
print({a:round(b) for a,b in d.get('kernel_ms_last_step',{}).items()})
"
