#!/bin/bash
# round 6: do the copy kernels of the result transfers (__amd_rocclr_copyBuffer: 108 ms of kernel time per step beside the stages) cost the stages anything?  The step with the DMA engines forced on / off
TAG=${1:-r06zd}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-22s %.3f s/step %.2f Gbp/s parity %s | %s" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/b0.json 2> $O/b0.err; line $O/b0.json default
( HSA_ENABLE_SDMA=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/b1.json 2> $O/b1.err; line $O/b1.json HSA_ENABLE_SDMA=1
( HSA_ENABLE_SDMA=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/b2.json 2> $O/b2.err; line $O/b2.json HSA_ENABLE_SDMA=0
( GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/b3.json 2> $O/b3.err; line $O/b3.json GPU_MAX_HW_QUEUES=8
