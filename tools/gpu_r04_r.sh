#!/bin/bash
# dmo after the address-space split of the denoise body: bench (md5 checked) + the dmo parity tests
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04r}; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo > $O/dmo.json 2> $O/dmo.err; python3 -c "
import json;d=json.loads(open('$O/dmo.json').read().strip().split('\n')[-1]);print('dmo %.3f s/step %.2f Gbp/s parity %s pairs-kernel %.0f ms'%(d['ms_per_step']/1e3,d['value'],d['parity'].get('match'),d['kernel_ms_last_step']['pairs']))"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "dmo" ) > $O/pytest_dmo.log 2>&1; tail -4 $O/pytest_dmo.log | head -2
