#!/bin/bash
# round 5: CIGAR text copied out beside the next range (wtz_fetch_cigar_text_begin / _end), switchable forms incl. the new ones, kernel statistics of one step, configs[3] shape
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05w}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "not switchable" > $O/pytest_forms.txt 2>&1; tail -2 $O/pytest_forms.txt
run(){ tag=$1; shift; args=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "kernel ms" $O/bench_$tag.err | tail -1
  grep "wall seconds" $O/bench_$tag.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().split('\n')[-1])
print('$tag', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'), d['config'].get('scratch'))
"
}
run zmo "" WTZ_X=0
run ecoli "--workload ecoli" WTZ_X=0
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 0 > $O/trace_zmo.log 2>&1 )
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace_zmo/**/*kernel_stats.csv',recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:28]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
( time timeout 2400 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err
grep "records,\|host seconds\|kernel ms\|wall seconds\|real\|z-mer index\|batches in\|splitting\|failed\|error" $O/bench_fly70.err | tail -14
python3 -c "
import json
d=json.loads(open('$O/bench_fly70.json').read().strip().split('\n')[-1])
print('fly70', d['ms_per_step'], d['value'], d.get('parity'))
"
rm -f /tmp/wtz_bench/reads_G140000000_*
