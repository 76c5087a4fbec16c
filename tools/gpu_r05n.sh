#!/bin/bash
# round 5: the window merge on the whole wavefront (wtz_merge_windows_wave): golden parity (zmo, wtgbo with its window step 0), configs[2] / configs[1] with md5
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05n}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --engine zmo --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_zmo.json 2> $O/bench_zmo.err
grep "kernel ms" $O/bench_zmo.err | tail -1
python3 -c "
import json
d=json.loads(open('$O/bench_zmo.json').read().strip().split('\n')[-1])
print('zmo', d['ms_per_step'], d['value'], 'K_pair', d['roofline_zmer']['kernel_ms_per_step'], d.get('parity',{}).get('match'))
"
timeout 400 python bench.py --workload ecoli --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; python3 -c "
import json
d=json.loads(open('$O/bench_ecoli.json').read().strip().split('\n')[-1])
print('ecoli', d['ms_per_step'], d['value'], d.get('parity',{}).get('match'))
"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_wtgbo.py tests/test_wtext.py -m gpu -x -q > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
