#!/bin/bash
# round 5: K_pair after the wave-parallel merge: phase profile, and the occupancy it is pinned to (3 / 4 / 5 waves per SIMD)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05o}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --engine zmo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; grep "kernel ms" $O/bench_base.err | tail -1
for v in occ3 occ5; do
  timeout 600 tools/with_variant.sh $v python bench.py --engine zmo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  echo "== $v"; grep "kernel ms" $O/bench_$v.err | tail -1
done
bash tools/gpu_phase_profile.sh $T
