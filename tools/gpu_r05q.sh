#!/bin/bash
# round 5: is the K-sw3 launch inside the step tail-bound?  The isolated kernel at launch sizes from 4 000 to 40 000 jobs, and the step with the four-wave kernel on the longest jobs
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05q}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
for n in 4000 8000 17000 40000; do
  timeout 300 python tools/ubench/ksw3_bench.py --forms 0 --jobs $n --reps 3 --no-compare 2>/dev/null | tail -1 | tee -a $O/ksw3_sizes.txt
done
run(){ tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['kernel_ms_last_step'])
"
}
run base WTZ_X=0
run mw2048 WTZ_SW_MW_MIN=2048
run mw1024 WTZ_SW_MW_MIN=1024
WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/bench_prof.json 2> $O/bench_prof.err
grep "ext-profile\] [0-9]* jobs" $O/bench_prof.err > $O/ext_launches.txt; wc -l $O/ext_launches.txt
