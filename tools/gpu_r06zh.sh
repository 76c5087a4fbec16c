#!/bin/bash
# round 6: K_gap (K-sw2 on a wavefront) compiled for three wavefronts per SIMD (-DWTZ_OCC_GAP=3: 168 registers, 34 spilled) against the default (213 registers, two waves)
TAG=${1:-r06zh}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-14s %.3f s/step %.2f Gbp/s parity %s | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
for i in 1 2; do ( timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline ) > $O/b$i.json 2> $O/b$i.err; line $O/b$i.json run$i; done
