#!/bin/bash
# round 6: partitioned second sketch level (phase clock at the configs[3] shape + the plain line), commit with the release store
TAG=${1:-r06j}
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
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err; line $O/bench_fly70.json fly70; grep "kernel ms" $O/bench_fly70.err | tail -1
( time WTZ_PROFILE_PAIR=1 timeout 2400 tools/with_variant.sh cprof python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline --no-verify ) > $O/cand_fly70.json 2> $O/cand_fly70.err
grep "cand-profile" $O/cand_fly70.err | python3 -c "
import sys,re
S=[0.0]*16
for l in sys.stdin:
    for m in re.finditer(r' (\d+):([0-9.]+)', l): S[int(m.group(1))]+=float(m.group(2))
print('fly70 cand slots', ' '.join('%d:%.1f' % (k, v) for k, v in enumerate(S)))"
grep "kernel ms" $O/cand_fly70.err | tail -1
rm -f /tmp/wtz_bench/reads_G140000000_*
