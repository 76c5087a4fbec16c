#!/bin/bash
# round 5, after the profile refresh: the driver's command again (its line now reads the PMC summary of THIS build's code objects) and the configs[4] shape on two ranks
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05lines}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 -c "
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().split('\n')[-1])
print('driver', d['ms_per_step'], d['value'], d['parity']['match'], d['roofline']['frac'], d['roofline'].get('traffic'), d['traffic_source'])
"
( time WTZ_BENCH_BACKEND=gloo timeout 1100 python bench.py --gpus 2 --workload human30 --steps 1 --warmup 0 --no-cpu-baseline --pool-gb 48 ) > $O/bench_human30_2ranks.json 2> $O/bench_human30_2ranks.err
python3 -c "
import json
d=json.loads(open('$O/bench_human30_2ranks.json').read().strip().split('\n')[-1])
print('human30', d['ms_per_step'], d['value'], d.get('parity'))
"
grep "real\|records,\|kernel ms" $O/bench_human30_2ranks.err | tail -4
