#!/bin/bash
# round 6: packed K-sw3 with the raised -10000 family for large init scores: DP forms, 40 000 jobs, goldens, the step
TAG=${1:-r06o}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s n %d %.3f s/step %.2f Gbp/s parity %s frac %.4f seed %.4f | %s | host %s" % (sys.argv[2], d['n_gpus'], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('roofline_seed') or {}).get('frac') or 0, {a:round(b) for a,b in k.items()}, d.get('host_seconds_last_step')))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time timeout 900 python -m pytest tests/test_gpu_dp_forms.py -m gpu -x -q -k "shift" ) > $O/pytest_forms.log 2>&1; tail -3 $O/pytest_forms.log
( time timeout 1200 python tools/ubench/ksw3_bench.py --forms 1,5,7 --reps 3 ) > $O/ksw3_bench.json 2> $O/ksw3_bench.err; cat $O/ksw3_bench.json
( time WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "left to the 32-bit" $O/bench_zmo.err | tail -1
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
( time timeout 600 python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json dmo
