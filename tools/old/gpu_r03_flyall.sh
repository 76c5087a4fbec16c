#!/bin/bash
# configs[3]-shape whole job with the ALL-reads z-mer index (160 GB) beside a 64 GB scratch pool instead of the per-batch index
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r03flya}; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time WTZ_TEST_KEEP_FLY=1 timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k fly ) > $O/pytest_fly.log 2>&1; grep -E "passed|failed" $O/pytest_fly.log
FLY=$(ls /tmp/wtz_bench/reads_G140000000_c70_s53.fa 2>/dev/null)
( time timeout 1200 bin/wtzmo -i $FLY -fo /tmp/fly_all.ovl -C -k 16 -s 200 -m 0.6 --zindex-batch 0 --pool-gb 64 ) > $O/fly_all.log 2>&1
grep -E "records,|kernel ms|real|batches in|failed|memory" $O/fly_all.log | cut -c1-260; md5sum /tmp/fly_all.ovl | cut -c1-32; rm -f /tmp/fly_all.ovl
rm -f /tmp/wtz_bench/reads_G140000000_c70_s53.fa*
