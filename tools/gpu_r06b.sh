#!/bin/bash
# round 6, second GPU run: the K-sw3 row diet + 2-column classes (DP vectors, 40 000 isolated jobs), the wave chain / z-read buckets / pool changes (golden parity, configs[2] md5),
# commit sections, and the kernel trace behind the WTZ_EXT_FR_SPLIT question (two launches vs three class launches per side)
TAG=${1:-r06b}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s %.3f s/step %.2f Gbp/s parity %s frac %.4f | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time timeout 600 python -m pytest tests/test_gpu_dp_forms.py -m gpu -x -q ) > $O/pytest_dp.log 2>&1; tail -3 $O/pytest_dp.log
timeout 900 python tools/ubench/ksw3_bench.py --forms 1,5,0 > $O/ksw3.txt 2> $O/ksw3.err; cat $O/ksw3.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 ) > $O/pytest_parity.log 2>&1; tail -14 $O/pytest_parity.log
( time timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "commit sections\|host seconds" $O/bench_zmo.err | tail -2
( time timeout 600 python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json dmo
cd /tmp
for v in nofuse split; do
  case $v in nofuse) E="WTZ_EXT_FUSED=0";; split) E="WTZ_EXT_FR_SPLIT=1";; esac
  env $E timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o t -- python $R/bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 1 > $O/trace_$v.log 2>&1
  python3 $R/tools/analysis/ext_launch_overlap.py $O/trace_$v > $O/ext_overlap_$v.txt 2>&1; tail -1 $O/ext_overlap_$v.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
