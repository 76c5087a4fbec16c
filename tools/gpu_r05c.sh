#!/bin/bash
# round 5: what bounds K-sw3?  The isolated bench (longest job first, one launch) on diagnostic builds of the library: no trace stores, no traceback,
# one row per job (the fixed cost of a job), one / three / four waves per SIMD
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base notrace notb notrtb rows1 occ1 occ3 occ4; do
  echo "== $v" >> $O/variants.txt
  timeout 300 tools/with_variant.sh $v python tools/ubench/ksw3_bench.py --forms 1,5,2 --no-compare --reps 2 >> $O/variants.txt 2>> $O/variants.err
done
cat $O/variants.txt
