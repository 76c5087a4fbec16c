#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
( time bin/wtzmo --repeat 2 --gpu-list 0,0,0 --pool-gb 48 -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) > /tmp/e3.log 2>&1; echo "rc=$?"; tail -12 /tmp/e3.log | cut -c1-300; md5sum /tmp/x.ovl
( time bin/wtzmo --gpu-list 0,0,0 --pool-gb 48 -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) > /tmp/e3b.log 2>&1; echo "rc=$?"; tail -6 /tmp/e3b.log | cut -c1-300; md5sum /tmp/x.ovl
