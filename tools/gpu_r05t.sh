#!/bin/bash
# round 5: new range defaults (fill 1.3, batches of 8 192), z-index built in chunks of reads, all-reads z-index at the configs[3] shape (pool sized from the input)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05t}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dmo" > $O/pytest_parity_dmo.txt 2>&1; tail -2 $O/pytest_parity_dmo.txt
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  grep -i "splitting\|exhaust" $O/bench_$tag.err | tail -2
  grep "batches in" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run zmo_onechunk "" WTZ_ZCHUNK_M=4000
run dmo "--engine dmo" WTZ_X=0
run ecoli "--workload ecoli" WTZ_X=0
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|host seconds\|kernel ms\|wall seconds\|real\|z-mer index\|batches in\|splitting" $O/bench_fly70.err | tail -12
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
