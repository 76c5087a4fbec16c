#!/bin/bash
# round 6, fourth GPU run: the frame kernel on four wavefronts for the longest items (DP form 6, isolated, in the step with three thresholds, kernel trace), K_zread without buckets, commit timers in parts, whole suite
TAG=${1:-r06d}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s n %d %.3f s/step %.2f Gbp/s parity %s frac %.4f seed %.4f msgs %s | %s | commit %s" % (sys.argv[2], d['n_gpus'], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('roofline_seed') or {}).get('frac') or 0, d.get('exchange_messages_per_step'), {a:round(b) for a,b in k.items()}, (d.get('host_seconds_last_step') or {}).get('commit')))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time timeout 600 python -m pytest tests/test_gpu_dp_forms.py -m gpu -x -q ) > $O/pytest_dp.log 2>&1; tail -3 $O/pytest_dp.log
timeout 900 python tools/ubench/ksw3_bench.py --forms 1,5,6,0 > $O/ksw3.txt 2> $O/ksw3.err; cat $O/ksw3.txt
for v in default off r1024 r4096; do
  case $v in default) E="";; off) E="WTZ_EXT_MW_ROWS=0";; r1024) E="WTZ_EXT_MW_ROWS=1024";; r4096) E="WTZ_EXT_MW_ROWS=4096";; esac
  ( time env $E WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_$v.json 2> $O/bench_$v.err; line $O/bench_$v.json $v
done
grep "commit sections\|in parts\|four wavefronts" $O/bench_default.err | tail -4
( time timeout 600 python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json dmo
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1; tail -20 $O/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_default -o t -- python $R/bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 1 > $O/trace_default.log 2>&1
python3 $R/tools/analysis/ext_launch_overlap.py $O/trace_default > $O/ext_overlap_default.txt 2>&1; tail -1 $O/ext_overlap_default.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
