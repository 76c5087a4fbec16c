#!/bin/bash
# dmo: LDS slice of K_pair_dm (20 KB product build vs 24 / 28 / 32 KB variants), two bench steps each
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04j}; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo"
show(){ python3 -c "
import json;d=json.loads(open('$O/$1.json').read().strip().split('\n')[-1]);print('$1 %.3f s/step %.2f Gbp/s parity %s pairs-kernel %.0f ms'%(d['ms_per_step']/1e3,d['value'],d['parity'].get('match'),d['kernel_ms_last_step']['pairs']))"; }
$B > $O/dm24.json 2> $O/dm24.err; show dm24
for kb in 20 22 28; do tools/with_variant.sh dm$kb $B > $O/dm$kb.json 2> $O/dm$kb.err; show dm$kb; done
