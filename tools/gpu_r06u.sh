#!/bin/bash
# round 6: what keeps the fused packed launch at two wavefronts per SIMD: scratch limit of the runtime? (HSA_SCRATCH_SINGLE_LIMIT), and the unfused launches (no scratch)
TAG=${1:-r06u}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-14s %.3f s/step %.2f Gbp/s parity %s frac %.4f | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( HSA_SCRATCH_SINGLE_LIMIT=2000000000 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/b1.json 2> $O/b1.err; line $O/b1.json scratch_limit
( WTZ_EXT_FUSED=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/b2.json 2> $O/b2.err; line $O/b2.json unfused_pk
( WTZ_EXT_FUSED=0 WTZ_EXT_PK=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/b3.json 2> $O/b3.err; line $O/b3.json unfused_fr
