#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
O=$R/gpurun_out/r04s6; mkdir -p $O
export TMPDIR=/tmp
show(){ python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().split('\n')[-1]); k=d['kernel_ms_last_step']; print('%-10s %.3f s/step parity %s pairs %.0f cand %.0f winalign %.0f stitch %.0f (ksw3 %.0f ksw2 %.0f)' % ('$2', d['ms_per_step']/1e3, d['parity'].get('match'), k['pairs'], k['candidates'], k['winalign'], k['stitch'], k['ksw3_wave'], k['ksw2_gap']))"; }
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_zmo.json 2> $O/bench_zmo.err; show $O/bench_zmo.json zmo
timeout 600 python bench.py --engine dmo --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_dmo.json 2> $O/bench_dmo.err; show $O/bench_dmo.json dmo
