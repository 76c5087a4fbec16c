#!/bin/bash
# round 5: batch overlap only where the batch's own record says few masks are being found (queries used / committed >= 0.85)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05ab}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "batches in" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run zmo_min70 "" WTZ_BATCH_OVERLAP_MIN_USED=0.7
run zmo_min93 "" WTZ_BATCH_OVERLAP_MIN_USED=0.93
run ecoli "--workload ecoli" WTZ_X=0
run ecoli_dmo "--workload ecoli --engine dmo" WTZ_X=0
run dmo "--engine dmo" WTZ_X=0
