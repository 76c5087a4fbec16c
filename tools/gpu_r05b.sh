#!/bin/bash
# round 5: the frame form of K-sw3 (wtz_sw_frame.h) - DP vectors x forms, the isolated K-sw3 bench, the whole step with md5 parity
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05b}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_dp_forms.py -x -q -k "shift" > $O/pytest_shift.txt 2>&1; tail -5 $O/pytest_shift.txt
timeout 900 python tools/ubench/ksw3_bench.py --forms 1,5,2,0 > $O/ksw3_bench.txt 2> $O/ksw3_bench.err; cat $O/ksw3_bench.txt; tail -3 $O/ksw3_bench.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fr.json 2> $O/bench_fr.err; tail -c 600 $O/bench_fr.err
WTZ_SW_MW_MIN=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fr_nomw.json 2> $O/bench_fr_nomw.err
WTZ_EXT_FR=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
for f in bench_fr bench_fr_nomw bench_old; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1])
print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('parity'))
"; done
