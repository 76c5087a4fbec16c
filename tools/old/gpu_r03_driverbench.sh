#!/bin/bash
# EXACTLY the driver's bench command line (r02's BENCH died at --steps 20 --warmup 5), timed, with the disk footprint watched beside it
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
df -h /tmp > $O/df_before.txt
( while true; do du -sm /tmp/wtz_bench 2>/dev/null | cut -f1; sleep 5; done ) > $O/du_mb.txt &
DU=$!
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
echo "rc=$?" | tee $O/bench_driver.rc
kill $DU
tail -1 $O/bench_driver.json | cut -c1-600
grep -E "real|\[bench\]" $O/bench_driver.err
echo "max du MB: $(sort -n $O/du_mb.txt | tail -1)"
df -h /tmp > $O/df_after.txt
