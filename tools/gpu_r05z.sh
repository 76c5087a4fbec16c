#!/bin/bash
# round 5: K_zread with 32-base walk pieces and seven LDS classes: a golden subset, configs[2] zmo, kernel statistics
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05z}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden" > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_zmo.json 2> $O/bench_zmo.err
grep "kernel ms" $O/bench_zmo.err | tail -1
python3 -c "
import json
d=json.loads(open('$O/bench_zmo.json').read().strip().split('\n')[-1])
print('zmo', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 0 > $O/trace_zmo.log 2>&1 )
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace_zmo/**/*kernel_stats.csv',recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:12]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
f=glob.glob('$O/trace_zmo/**/*kernel_trace.csv',recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'K_zread' in r['Kernel_Name']: print('K_zread grid', r['Grid_Size_X'], 'wg', r['Workgroup_Size_X'], 'lds', r['LDS_Block_Size'], 'ms', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
