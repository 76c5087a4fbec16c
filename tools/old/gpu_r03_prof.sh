#!/bin/bash
# device phase profiler (library built with -DWTZ_PROFILE: tools/exp_libs/libwtzmo_hip_PROFILE.so) on one configs[2] step per engine
TAG=${1:-r03prof}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
FA=$(ls /tmp/wtz_bench/reads_G12000000_c100_s29.fa)
cp smartdenovo_amd/libwtzmo_hip.so /tmp/lib_orig.so
cp tools/exp_libs/libwtzmo_hip_PROFILE.so smartdenovo_amd/libwtzmo_hip.so
WTZ_PROFILE_PAIR=1 timeout 600 bin/wtzmo -i $FA -fo /tmp/wtz_bench/prof.ovl -k 16 -s 200 -m 0.6 2> $O/prof_zmo.err
grep -E "phase-profile|kernel ms" $O/prof_zmo.err | tail -3 | cut -c1-1500
grep -E "pair-profile" $O/prof_zmo.err | tail -3 | cut -c1-400
cp /tmp/lib_orig.so smartdenovo_amd/libwtzmo_hip.so; rm -f /tmp/wtz_bench/prof.ovl*
