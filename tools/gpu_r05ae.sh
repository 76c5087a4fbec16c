#!/bin/bash
# round 5: seed lookup scratch (group list in place of the dead tuple list), batch cap from the measured scratch per query: parity subset, configs[2], configs[3] shape
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05ae}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden" > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run dmo "--engine dmo" WTZ_X=0
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|host seconds\|kernel ms\|batches in\|splitting\|failed\|error" $O/bench_fly70.err | tail -8
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
