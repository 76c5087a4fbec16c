#!/bin/bash
# round 6: full GPU suite and the configs[3]-shape line on the packed K-sw3
TAG=${1:-r06zb}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-14s %.3f s/step %.2f Gbp/s parity %s frac %.4f seed %.4f | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('roofline_seed') or {}).get('frac') or 0, {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time WTZ_TEST_NO_FLY=1 timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time WTZ_PROFILE_PAIR=1 timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err; line $O/bench_fly70.json fly70; grep "dealt" $O/bench_fly70.err | tail -1
rm -f /tmp/wtz_bench/reads_G140000000_*
