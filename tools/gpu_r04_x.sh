#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
O=$R/gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
show(){ python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().split('\n')[-1]); k=d['kernel_ms_last_step']; print('%-10s %.3f s/step parity %s pairs %.0f cand %.0f winalign %.0f stitch %.0f (ksw3 %.0f ksw2 %.0f)' % ('$2', d['ms_per_step']/1e3, d['parity'].get('match'), k['pairs'], k['candidates'], k['winalign'], k['stitch'], k['ksw3_wave'], k['ksw2_gap']))"; }
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_zmo.json 2> $O/bench_zmo.err; show $O/bench_zmo.json zmo
WTZ_PAIR_HEAVY_FIRST=1 timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_zmo_hf.json 2> $O/bench_zmo_hf.err; show $O/bench_zmo_hf.json zmo_heavyfirst
timeout 600 python bench.py --engine dmo --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_dmo.json 2> $O/bench_dmo.err; show $O/bench_dmo.json dmo
WTZ_PROFILE_PAIR=1 timeout 600 tools/with_variant.sh prof python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine zmo > $O/slots_zmo.json 2> $O/slots_zmo.err
grep "phase-profile" $O/slots_zmo.err | tr ' ' '\n' | grep -v ":0.0$" | tr '\n' ' '; echo
grep "pair-profile\] n=" $O/slots_zmo.err | head -4
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log | head -n 2
