#!/bin/bash
# f1 / f4 on the GPU box: function-level align_hzmaux parity, device ingest parity + rate, a short bench line carrying roofline_ingest
TAG=${1:-r03f}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1200 python -m pytest tests/test_hzmaux_functions.py tests/test_ingest.py -x -q -m gpu -s ) > $O/pytest.log 2>&1; grep -E "passed|failed|ingest " $O/pytest.log | tail -8
( time python3 bench.py --gpus 1 --steps 3 --warmup 1 ) > $O/bench_zmo.json 2> $O/bench_zmo.err; tail -1 $O/bench_zmo.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline_ingest'))"
grep "packed on the device" $O/bench_zmo.err
