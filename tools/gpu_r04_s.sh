#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04s}; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log | head -2
for e in zmo dmo; do timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine $e > $O/bench_$e.json 2> $O/bench_$e.err; python3 -c "
import json;d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1]);print('$e %.3f s/step %.2f Gbp/s parity %s kernels %s'%(d['ms_per_step']/1e3,d['value'],d['parity'].get('match'),{k:round(v) for k,v in d['kernel_ms_last_step'].items()}))"; done
