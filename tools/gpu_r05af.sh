#!/bin/bash
# round 5: range budget scaled by the measured pairs per candidate row (host only); configs[2] both engines, configs[1], configs[3] shape, then the driver's command
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05af}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1; grep "batches in" $O/bench_$tag.err | tail -1; grep -c "splitting" $O/bench_$tag.err
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run zmo_f13 "" WTZ_RANGE_FILL=1.3
run dmo "--engine dmo" WTZ_X=0
run ecoli "--workload ecoli" WTZ_X=0
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|kernel ms\|batches in\|failed\|error" $O/bench_fly70.err | tail -5; grep -c "splitting" $O/bench_fly70.err
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
