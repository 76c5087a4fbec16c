#!/bin/bash
# z-mer prefilter (both engines) + dmo band walk / seeds: parity of both engines, bench lines, dmo phase profile
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04g}; mkdir -p $O
export TMPDIR=/tmp WTZ_TEST_NO_FLY=1
cd $R
( time timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_hzmaux_functions.py tests/test_wtgbo.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for e in zmo dmo; do python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine $e > $O/bench_$e.json 2> $O/bench_$e.err; python3 -c "
import json;d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1]);print('$e %.3f s/step %.2f Gbp/s parity %s kernels %s'%(d['ms_per_step']/1e3,d['value'],d['parity'].get('match'),{k:round(v) for k,v in d['kernel_ms_last_step'].items()}))"; done
for e in dmo zmo; do
WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine $e > $O/slots_$e.json 2> $O/slots_$e.err
grep "phase-profile" $O/slots_$e.err | grep -v " 2:0.0 .* 17:0.0" | tr ' ' '\n' | grep -E "^(1[2-9]|2[0-9]|3[01]|4[6-9]|5[0-8]|11|2|6|7):" | tr '\n' ' '; echo
grep "pair-profile\] n=" $O/slots_$e.err | python3 -c "
import sys,re
S=[0]*4;n=0
for l in sys.stdin:
    m=re.search(r'n=(\d+) kticks sum match/sort/win/total (\d+)/(\d+)/(\d+)/(\d+)',l)
    if m: n+=int(m.group(1)); S=[S[k]+int(m.group(2+k)) for k in range(4)]
print('$e pairs',n,'Gticks match/sort/win/total',[round(x*1024/1e9) for x in S])"
done
