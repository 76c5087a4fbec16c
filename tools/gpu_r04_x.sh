#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
O=$R/gpurun_out/r04z; mkdir -p $O
export TMPDIR=/tmp
show(){ python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().split('\n')[-1]); k=d['kernel_ms_last_step']; print('%-10s %.3f s/step parity %s pairs %.0f winalign %.0f stitch %.0f (ksw3 %.0f ksw2 %.0f)' % ('$2', d['ms_per_step']/1e3, d['parity'].get('match'), k['pairs'], k['winalign'], k['stitch'], k['ksw3_wave'], k['ksw2_gap']))"; }
timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; show $O/bench_full.json full
for v in nocache nojoin ab; do
  [ -f smartdenovo_amd/variants/libwtzmo_hip_$v.so ] || continue
  timeout 600 tools/with_variant.sh $v python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err; show $O/bench_$v.json $v
done
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_wtext.py tests/test_wtgbo.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log | head -n 2
