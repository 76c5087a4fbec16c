#!/bin/bash
# round 6, third GPU run: K-sw3 row diet + 2-column classes, 4-bit sketch, packed rank messages, test pool default: DP vectors, isolated K-sw3, whole suite (durations), bench lines, 2-rank message counts
TAG=${1:-r06c}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s n %d %.3f s/step %.2f Gbp/s parity %s frac %.4f seed %.4f msgs %s | %s | host %s" % (sys.argv[2], d['n_gpus'], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('roofline_seed') or {}).get('frac') or 0, d.get('exchange_messages_per_step'), {a:round(b) for a,b in k.items()}, d.get('host_seconds_last_step')))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time timeout 600 python -m pytest tests/test_gpu_dp_forms.py -m gpu -x -q ) > $O/pytest_dp.log 2>&1; tail -3 $O/pytest_dp.log
timeout 900 python tools/ubench/ksw3_bench.py --forms 1,5,0 > $O/ksw3.txt 2> $O/ksw3.err; cat $O/ksw3.txt
( time timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "commit sections" $O/bench_zmo.err | tail -1
( time timeout 600 python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json dmo
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; tail -22 $O/pytest.log
WTZ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload ecoli --no-cpu-baseline > $O/bench_ecoli_2ranks_gloo.json 2> $O/bench_ecoli_2ranks_gloo.err; line $O/bench_ecoli_2ranks_gloo.json "2r-ecoli"
WTZ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 2 --warmup 1 --pool-gb 48 > $O/bench_yeast_2ranks_gloo.json 2> $O/bench_yeast_2ranks_gloo.err; line $O/bench_yeast_2ranks_gloo.json "2r-yeast"
