#!/bin/bash
# round 5: block merge / chain rewritten (chain on the whole wave), frame kernel per band class (WTZ_EXT_FR_SPLIT): DP vectors, K-sw3 alone, dmo parity + both engines with md5, phase profile
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05j}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_dp_forms.py -x -q > $O/pytest_dp.txt 2>&1; tail -2 $O/pytest_dp.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dmo" > $O/pytest_parity_dmo.txt 2>&1; tail -2 $O/pytest_parity_dmo.txt
for sp in 0 1; do
  echo "== WTZ_EXT_FR_SPLIT=$sp" >> $O/ksw3_bench.txt
  WTZ_EXT_FR_SPLIT=$sp timeout 300 python tools/ubench/ksw3_bench.py --forms 1,0 --reps 3 >> $O/ksw3_bench.txt 2>> $O/ksw3_bench.err
done
cat $O/ksw3_bench.txt
for sp in 0 1; do
  WTZ_EXT_FR_SPLIT=$sp timeout 600 python bench.py --engine zmo --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_zmo_split$sp.json 2> $O/bench_zmo_split$sp.err
  grep "kernel ms" $O/bench_zmo_split$sp.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_zmo_split$sp.json').read().strip().split('\n')[-1])
print('zmo split$sp', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('parity',{}).get('match'))
"
done
timeout 600 python bench.py --engine dmo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dmo.json 2> $O/bench_dmo.err
grep "kernel ms" $O/bench_dmo.err | tail -1
python3 -c "
import json
d=json.loads(open('$O/bench_dmo.json').read().strip().split('\n')[-1])
print('dmo', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
bash tools/gpu_phase_profile.sh $T
