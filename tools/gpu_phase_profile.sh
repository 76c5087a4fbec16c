#!/bin/bash
# Where the pair kernels spend their wave time: the phase-profiler build (-DWTZ_PROFILE: clock ticks per slot, accumulated in LDS by lane 0 of every task) run for one
# step of each engine.  Build the variant first (no GPU needed):  tools/build_variant.sh prof -DWTZ_PROFILE
# usage (GPU box): tools/gpu_phase_profile.sh <tag>    -> gpurun_out/<tag>/slots_{dmo,zmo}.err; slot numbers are named where WTZ_PROF_ADD / WTZ_PROF_CNT use them
# (dmo denoise: wtz_dotmatrix.h; zmo window scans: wtz_window.h 12, 16-31; merge loop incl. scans 63 and chain 62: wtz_tasks.h; K-sw3 jobs: wtz_sw_wave.h)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-phase}; mkdir -p $O
export TMPDIR=/tmp
cd $R
for e in dmo zmo; do
  WTZ_PROFILE_PAIR=1 timeout 600 tools/with_variant.sh prof python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine $e > $O/slots_$e.json 2> $O/slots_$e.err
  grep "phase-profile" $O/slots_$e.err | tr ' ' '\n' | grep -v ":0.0$" | tr '\n' ' '; echo
  grep "pair-profile\] n=" $O/slots_$e.err | python3 -c "
import sys,re
S=[0]*4;n=0
for l in sys.stdin:
    m=re.search(r'n=(\d+) kticks sum match/sort/win/total (\d+)/(\d+)/(\d+)/(\d+)',l)
    if m: n+=int(m.group(1)); S=[S[k]+int(m.group(2+k)) for k in range(4)]
print('$e pairs',n,'Gticks match/sort/windows-or-denoise/total',[round(x*1024/1e9) for x in S])"
done
