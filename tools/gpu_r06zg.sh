#!/bin/bash
# round 6: left extensions of the longest items started beside the K-sw2 stage (WTZ_EXT_EARLY=<rows>): the step with it off / at 1500 / at 800 rows, goldens
TAG=${1:-r06zg}
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
for e in 0 1500 800 1500 0; do
( WTZ_PROFILE_PAIR=1 WTZ_EXT_EARLY=$e timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline ) > $O/b_$e.json 2> $O/b_$e.err; line $O/b_$e.json early_$e; grep "dealt" $O/b_$e.err | tail -1 | cut -c1-260
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
