#!/bin/bash
# round 5: range size sweep with the fused extension launch: WTZ_RANGE_FILL x --batch x --pool-gb at configs[2] (zmo), then dmo at the chosen fill
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05s}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  grep -i "splitting\|exhaust" $O/bench_$tag.err | tail -2
  grep "batches in" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run f18 "" WTZ_RANGE_FILL=1.8
run f14_b8k "--max-batch 8192" WTZ_RANGE_FILL=1.4
run f14_p200 "--pool-gb 200" WTZ_RANGE_FILL=1.4
run f14_p200_b8k "--pool-gb 200 --max-batch 8192" WTZ_RANGE_FILL=1.4
run dmo_f07 "--engine dmo" WTZ_RANGE_FILL=0.7
run dmo_f14 "--engine dmo" WTZ_RANGE_FILL=1.4
