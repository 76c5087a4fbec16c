#!/bin/bash
# round 5, last GPU action: the driver's command (its line reads the PMC summary of this build's code objects: bench.kernel_source_id over .text / .rodata / .note)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05last}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 -c "
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().split('\n')[-1])
print('driver', d['ms_per_step'], d['value'], d['parity']['match'], d['roofline']['frac'], d['build'], d['traffic_source'])
"
