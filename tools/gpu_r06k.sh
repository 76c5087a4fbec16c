#!/bin/bash
# round 6: second sketch level without gathers, retired K-sw3 forms: goldens, configs[2] line, configs[3] shape; commit timers (flush, helper control)
TAG=${1:-r06k}
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
( time timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "helper threads" $O/bench_zmo.err | tail -1
( time timeout 600 python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json dmo
( time WTZ_TEST_NO_FLY=1 timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err; line $O/bench_fly70.json fly70; grep "kernel ms\|helper threads" $O/bench_fly70.err | tail -2
rm -f /tmp/wtz_bench/reads_G140000000_*
