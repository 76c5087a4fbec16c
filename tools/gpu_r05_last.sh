#!/bin/bash
# round 5, last GPU action: the whole GPU suite on the final host code, then the driver's command (its line reads the PMC summary of this build's code objects)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05last}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp WTZ_TEST_KEEP_STDERR=$O/stderr
cd $R
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 -c "
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().split('\n')[-1])
print('driver', d['ms_per_step'], d['value'], d['parity']['match'], d['roofline']['frac'], d['traffic_source'])
"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | head -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
