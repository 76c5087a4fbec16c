#!/bin/bash
# K-sw3 placement experiments (configs[2] zmo): band classes in separate launches, wide bands on four waves
TAG=${1:-r03ext}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
for cfg in "WTZ_EXT_SPLIT=0" "WTZ_EXT_SPLIT=2" "WTZ_EXT_SPLIT=2 WTZ_EXT_MW_CW=20" "WTZ_EXT_SPLIT=2 WTZ_EXT_MW_CW=16" "WTZ_EXT_SPLIT=2 WTZ_EXT_MW_CW=12" "WTZ_EXT_SPLIT=2 WTZ_EXT_MW_CW=0" "WTZ_EXT_SPLIT=1 WTZ_EXT_MW_CW=20" "WTZ_EXT_SPLIT=0"; do
  n=$(echo $cfg | tr ' =' '__')
  env $cfg python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err
  echo "$cfg: $(tail -1 $O/bench_$n.json | cut -c88-180)"; grep "kernel ms" $O/bench_$n.err | tail -1 | cut -c1-200
done
