#!/bin/bash
# heavy-first task order of the K_pair launches: dmo on / off, zmo off / on (the bench checks the md5 of every run)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04m}; mkdir -p $O
export TMPDIR=/tmp
cd $R
show(){ python3 -c "
import json;d=json.loads(open('$O/$1.json').read().strip().split('\n')[-1]);print('$1 %.3f s/step %.2f Gbp/s parity %s pairs-kernel %.0f ms'%(d['ms_per_step']/1e3,d['value'],d['parity'].get('match'),d['kernel_ms_last_step']['pairs']))"; }
for hf in 1 0; do WTZ_PAIR_HEAVY_FIRST=$hf python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo > $O/dmo_hf$hf.json 2> $O/dmo_hf$hf.err; show dmo_hf$hf; done
for hf in 0 1; do WTZ_PAIR_HEAVY_FIRST=$hf python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/zmo_hf$hf.json 2> $O/zmo_hf$hf.err; show zmo_hf$hf; done
for kb in 20; do WTZ_PAIR_HEAVY_FIRST=1 tools/with_variant.sh dm$kb python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo > $O/dmo_dm$kb.json 2> $O/dmo_dm$kb.err; show dmo_dm$kb; done
