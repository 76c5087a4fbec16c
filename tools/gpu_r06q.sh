#!/bin/bash
# round 6: K-sw3 packed against the 32-bit frame form on few long jobs (latency of a lone wavefront) and on many
TAG=${1:-r06q}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
for n in 64 2000 10000; do
( timeout 600 python tools/ubench/ksw3_bench.py --forms 5,7 --reps 3 --jobs $n ) > $O/ksw3_bench_$n.json 2> $O/ksw3_bench_$n.err; cat $O/ksw3_bench_$n.json
done
