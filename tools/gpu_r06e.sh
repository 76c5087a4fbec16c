#!/bin/bash
# round 6, fifth GPU run: commit with prefetch + typed sorts (timers), the four-wave form in a parity run, single-job latency one wave vs four, and the configs[3] shape (seed lookup with the 4-bit sketch)
TAG=${1:-r06e}
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
( time timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "commit sections\|in parts" $O/bench_zmo.err | tail -2
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "EXT_MW_ROWS or EXT_FUSED or golden" ) > $O/pytest_forms.log 2>&1; tail -3 $O/pytest_forms.log
for n in 300 3000; do timeout 600 python tools/ubench/ksw3_bench.py --jobs $n --forms 5,6 > $O/ksw3_$n.txt 2> $O/ksw3_$n.err; cat $O/ksw3_$n.txt; done
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err; line $O/bench_fly70.json fly70; grep "records,\|host seconds\|kernel ms\|commit sections" $O/bench_fly70.err | tail -5
rm -f /tmp/wtz_bench/reads_G140000000_*
