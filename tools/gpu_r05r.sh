#!/bin/bash
# round 5: both end extensions of an item on one wavefront (wtz_stitch_fused.h) - golden parity, then the configs[2] step with and without, and larger ranges (WTZ_RANGE_FILL)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05r}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not dmo" > $O/pytest_parity_zmo.txt 2>&1; tail -3 $O/pytest_parity_zmo.txt
run(){ tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  grep -i "split\|exhaust" $O/bench_$tag.err | tail -2
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'), d.get('pairs_per_step'))
"
}
run fused WTZ_X=0
run unfused WTZ_EXT_FUSED=0
run fused_fill10 WTZ_RANGE_FILL=1.0
run fused_fill14 WTZ_RANGE_FILL=1.4
WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/bench_prof.json 2> $O/bench_prof.err
grep "ext-profile\] fused" $O/bench_prof.err > $O/ext_launches.txt; wc -l $O/ext_launches.txt
run fused_ctx2 WTZ_BENCH_CONTEXTS=2
