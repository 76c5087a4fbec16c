#!/bin/bash
# the three bench lines only (default zmo with CPU baseline, dmo, E. coli shape)
TAG=${1:-r02y}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time python bench.py ) > $O/bench_zmo.json 2> $O/bench_zmo.err; tail -1 $O/bench_zmo.json | cut -c1-400; grep real $O/bench_zmo.err
python bench.py --engine dmo > $O/bench_dmo.json 2> $O/bench_dmo.err; tail -1 $O/bench_dmo.json | cut -c1-300
python bench.py --workload ecoli --no-cpu-baseline > $O/bench_ecoli_zmo.json 2> $O/bench_ecoli_zmo.err; tail -1 $O/bench_ecoli_zmo.json | cut -c1-300
python bench.py --workload ecoli --engine dmo --no-cpu-baseline > $O/bench_ecoli_dmo.json 2> $O/bench_ecoli_dmo.err; tail -1 $O/bench_ecoli_dmo.json | cut -c1-300
