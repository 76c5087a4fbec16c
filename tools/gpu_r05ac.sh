#!/bin/bash
# round 5: batch overlap decided by the expected share of the next batch that the overlapped commit masks
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05ad}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp WTZ_BATCH_OVERLAP_TRACE=1
cd $R
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "batches in" $O/bench_$tag.err | tail -1
  grep "batch-overlap" $O/bench_$tag.err | tail -${NB:-6}
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run zmo_always "" WTZ_BATCH_OVERLAP_GAIN=1e9
run ecoli "--workload ecoli" WTZ_X=0
run ecoli_dmo "--workload ecoli --engine dmo" WTZ_X=0
NB=18 run dmo "--engine dmo" WTZ_X=0
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|batches in\|splitting\|failed\|error" $O/bench_fly70.err | tail -4; grep -c "started early" $O/bench_fly70.err; grep -c "formed only" $O/bench_fly70.err
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
