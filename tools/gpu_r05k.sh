#!/bin/bash
# round 5: why is the frame kernel per band class slower inside the step?  three modes (one launch / three streams / one stream), per-call lines of the extension stage
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05k}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
for sp in 0 1 2; do
  WTZ_PROFILE_PAIR=1 WTZ_EXT_FR_SPLIT=$sp timeout 600 python bench.py --engine zmo --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $O/bench_zmo_split$sp.json 2> $O/bench_zmo_split$sp.err
  grep "kernel ms" $O/bench_zmo_split$sp.err | tail -1
  grep "ext-profile\] [0-9]* jobs" $O/bench_zmo_split$sp.err | tail -12 | cut -c1-150
done
