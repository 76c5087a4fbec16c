#!/bin/bash
# round 6: lane-per-problem K-sw1 / K-sw2 with the row loop split by band shift: DP forms, goldens, the step
TAG=${1:-r06zc}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-14s %.3f s/step %.2f Gbp/s parity %s frac %.4f | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/b1.json 2> $O/b1.err; line $O/b1.json zmo
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp_forms.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
