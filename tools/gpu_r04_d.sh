#!/bin/bash
# phase-profiler build: where the dmo denoise and the zmo window scans spend their wave time
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04d}; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify"
WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof $B --engine dmo > $O/slots_dmo.json 2> $O/slots_dmo.err
grep "phase-profile" $O/slots_dmo.err | grep -v " 2:0.0" | cut -c1-1800
#WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof $B > $O/slots_zmo.json 2> $O/slots_zmo.err
#grep "phase-profile" $O/slots_zmo.err | grep -v " 17:0.0" | cut -c1-1800
