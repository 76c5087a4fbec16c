#!/bin/bash
# round 5: seed lookup, second step (run descriptors 64 at a time, survivors listed once, scatter from the list): product build and the 512-thread variant, both engines with md5; phase clock of both
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05i}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in product t512; do
for e in zmo dmo; do
  if [ $v = product ]; then timeout 600 python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$e.json 2> $O/bench_${v}_$e.err
  else timeout 600 tools/with_variant.sh $v python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$e.json 2> $O/bench_${v}_$e.err; fi
  python3 -c "
import json
d=json.loads(open('$O/bench_${v}_$e.json').read().strip().split('\n')[-1])
print('$v $e', d['ms_per_step'], d['value'], 'seed ms', d['roofline_seed']['kernel_ms_per_step'], d['roofline_seed']['frac'], d.get('parity',{}).get('match'))
"
done; done
bash tools/gpu_cand_profile.sh $T cprof
bash tools/gpu_cand_profile.sh ${T}_512 cprof512
