#!/bin/bash
# two / three contexts on ONE device (parts dealt round-robin, central commit) against one context with a large pool
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ echo "== $*"; ( time bin/wtzmo "$@" -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "kernel ms|real|batches in|host seconds|wall seconds|step" | cut -c1-260; md5sum /tmp/x.ovl | cut -c1-32; }
run --repeat 2 --pool-gb 128
run --repeat 2 --gpu-list 0,0 --pool-gb 64
run --repeat 2 --gpu-list 0,0,0 --pool-gb 48
