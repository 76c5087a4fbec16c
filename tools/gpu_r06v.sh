#!/bin/bash
# round 6: packed K-sw3 with the items outside its window dealt to a concurrent 32-bit launch: the step, kernel durations, goldens
TAG=${1:-r06v}
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
( WTZ_PROFILE_PAIR=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/b1.json 2> $O/b1.err; line $O/b1.json pk; grep "dealt" $O/b1.err | tail -1
( WTZ_EXT_FUSED=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/b2.json 2> $O/b2.err; line $O/b2.json unfused_pk
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/prof -o zmo -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $O/prof.log 2>&1; cd $R
python3 - $O/prof <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'stitch_ext' in r['Kernel_Name']]
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-24:]: print("%9.2f +%7.2f ms %8s %s q%s" % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r['Grid_Size_X'], r['Kernel_Name'][16:34], r['Queue_Id']))
PY
rm -rf $O/prof
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
