#!/bin/bash
TAG=${1:-r03xcd}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
for g in 0 256 64 1024 0 256; do
  WTZ_XCD_GROUP=$g python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_zmo_$g.json 2> $O/bench_zmo_$g.err
  echo "zmo xcd-group $g: $(tail -1 $O/bench_zmo_$g.json | cut -c88-190)"; grep -E "kernel ms" $O/bench_zmo_$g.err | tail -1 | cut -c1-140
done
for g in 0 256; do
  WTZ_XCD_GROUP=$g python bench.py --engine dmo --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_dmo_$g.json 2> $O/bench_dmo_$g.err
  echo "dmo xcd-group $g: $(tail -1 $O/bench_dmo_$g.json | cut -c88-190)"; grep -E "kernel ms" $O/bench_dmo_$g.err | tail -1 | cut -c1-140
done
