#!/bin/bash
# round 4: where K_pair (both engines) and the K-sw3 launches spend their time - per-pair tick sums, per-launch extension statistics, phase-profiler slots
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04c}; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify"
WTZ_PROFILE_PAIR=1 $B --engine dmo > $O/prof_dmo.json 2> $O/prof_dmo.err
grep "pair-profile" $O/prof_dmo.err | tail -6
WTZ_PROFILE_PAIR=1 $B > $O/prof_zmo.json 2> $O/prof_zmo.err
grep "pair-profile\|ext-profile\|lane-profile" $O/prof_zmo.err | tail -14
WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof $B > $O/slots_zmo.json 2> $O/slots_zmo.err
grep "phase-profile" $O/slots_zmo.err | tail -3 | cut -c1-1500
WTZ_PROFILE_PAIR=1 tools/with_variant.sh prof $B --engine dmo > $O/slots_dmo.json 2> $O/slots_dmo.err
grep "phase-profile" $O/slots_dmo.err | tail -2 | cut -c1-1500
