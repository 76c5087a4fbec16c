#!/bin/bash
# round 6: the packed 16-bit K-sw3 inside the step (fused launch): goldens, configs[2] zmo line with and without it, ecoli
TAG=${1:-r06n}
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
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -5 $O/pytest_parity.log
( time WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "left to the 32-bit" $O/bench_zmo.err | tail -1
( time WTZ_EXT_PK=0 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo_nopk.json 2> $O/bench_zmo_nopk.err; line $O/bench_zmo_nopk.json zmo_nopk
( time timeout 600 python bench.py --workload ecoli --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_ecoli.json 2> $O/bench_ecoli.err; line $O/bench_ecoli.json ecoli
