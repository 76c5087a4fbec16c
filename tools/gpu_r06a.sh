#!/bin/bash
# round 6, first GPU run: the suite with per-test durations (what to trim), today's baseline of the bench line, and the WTZ_EXT_FR_SPLIT question
# (fused launch vs two launches vs two launches split by band class, in the step and alone)
TAG=${1:-r06a}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in default nofuse split; do
  case $v in default) E="";; nofuse) E="WTZ_EXT_FUSED=0";; split) E="WTZ_EXT_FR_SPLIT=1";; esac
  ( time env $E WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s %.3f s/step %.2f Gbp/s parity %s frac %.4f | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
timeout 900 python tools/ubench/ksw3_bench.py --forms 1,5,0 > $O/ksw3_default.txt 2> $O/ksw3_default.err; cat $O/ksw3_default.txt
WTZ_EXT_FR_SPLIT=1 timeout 900 python tools/ubench/ksw3_bench.py --forms 1,0 > $O/ksw3_split.txt 2> $O/ksw3_split.err; cat $O/ksw3_split.txt
