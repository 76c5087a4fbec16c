#!/bin/bash
# configs[3]-shape whole job: queries per batch in the per-batch z-index mode (WTZ_ZBATCH_MAX = 512 (default) / 1024 / 2048)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r03flyb}; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time WTZ_TEST_KEEP_FLY=1 timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k fly ) > $O/pytest_fly.log 2>&1; tail -1 $O/pytest_fly.log
FLY=$(ls /tmp/wtz_bench/reads_G140000000_c70_s53.fa 2>/dev/null)
for zb in 2048 1024; do
  echo "== WTZ_ZBATCH_MAX=$zb"
  ( time WTZ_ZBATCH_MAX=$zb timeout 1200 bin/wtzmo -i $FLY -fo /tmp/fly_$zb.ovl -C -k 16 -s 200 -m 0.6 --batch 8192 ) > $O/fly_$zb.log 2>&1
  grep -E "records,|kernel ms|real|pool peak" $O/fly_$zb.log | cut -c1-260; md5sum /tmp/fly_$zb.ovl | cut -c1-32; rm -f /tmp/fly_$zb.ovl
done
rm -f /tmp/wtz_bench/reads_G140000000_c70_s53.fa*
