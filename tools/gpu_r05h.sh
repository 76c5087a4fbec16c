#!/bin/bash
# round 5: seed lookup with the group sketch, wave-per-k-mer seed walks and the parallel fold: golden parity on the GPU, both engines at configs[2] with md5, the phase clock
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05h}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
for e in zmo dmo; do
  timeout 600 python bench.py --engine $e --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
  grep "kernel ms" $O/bench_$e.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$e.json').read().strip().split('\n')[-1])
print('$e', d['ms_per_step'], d['value'], d['roofline_seed']['kernel_ms_per_step'], d['roofline_seed']['frac'], d.get('parity'))
"
done
[ -f smartdenovo_amd/variants/libwtzmo_hip_cprof.so ] && bash tools/gpu_cand_profile.sh $T cprof
