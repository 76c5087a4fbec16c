#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
bash tools/gpu_wtext_scale.sh r04wtext
O=$R/gpurun_out/r04wtext
WTZ_PROFILE_PAIR=1 timeout 600 tools/with_variant.sh prof python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine zmo > $O/slots_zmo.json 2> $O/slots_zmo.err
grep "phase-profile" $O/slots_zmo.err | tr ' ' '\n' | grep -v ":0.0$" | tr '\n' ' '; echo
