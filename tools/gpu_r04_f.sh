#!/bin/bash
# dmo denoise rework: parity (every dmo golden, the heavy-pair paths, configs[1] / [2] / repeat-rich md5), then the bench line and the phase profile
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04f}; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "dmo or switchable" ) > $O/pytest_dmo.log 2>&1; tail -4 $O/pytest_dmo.log
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo > $O/bench_dmo.json 2> $O/bench_dmo.err; python3 -c "
import json;d=json.loads(open('$O/bench_dmo.json').read().strip().split('\n')[-1]);print('dmo %.3f s/step %.2f Gbp/s parity %s pairs-kernel %.0f ms'%(d['ms_per_step']/1e3,d['value'],d['parity'],d['kernel_ms_last_step']['pairs']))"
WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine dmo > $O/slots_dmo.json 2> $O/slots_dmo.err
grep "phase-profile" $O/slots_dmo.err | grep -v " 2:0.0" | tr ' ' '\n' | grep -E "^(2[4-9]|3[01]|4[6-9]|5[0-9]|10|11|2|6|62):" | tr '\n' ' '; echo
grep "pair-profile\] n=" $O/slots_dmo.err | python3 -c "
import sys,re
S=[0]*4;n=0
for l in sys.stdin:
    m=re.search(r'n=(\d+) kticks sum match/sort/win/total (\d+)/(\d+)/(\d+)/(\d+)',l)
    if m: n+=int(m.group(1)); S=[S[k]+int(m.group(2+k)) for k in range(4)]
print('pairs',n,'Gticks match/sort/denoise+/total',[round(x*1024/1e9) for x in S])"
