#!/bin/bash
# round 6: K-sw3 with two 16-bit cells per register (wtz_sw_frame16.h, DP form 7): the reference's vectors, then 40 000 dumped jobs against forms 1 and 5
TAG=${1:-r06m}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_gpu_dp_forms.py -m gpu -x -q -k "shift" ) > $O/pytest_forms.log 2>&1; tail -15 $O/pytest_forms.log
( time timeout 1200 python tools/ubench/ksw3_bench.py --forms 1,5,7 --reps 3 ) > $O/ksw3_bench.json 2> $O/ksw3_bench.err; cat $O/ksw3_bench.json; tail -3 $O/ksw3_bench.err
