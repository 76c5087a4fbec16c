#!/bin/bash
# pool size / batch size sweep on configs[2] (zmo): fewer, larger ranges = fewer launch tails
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ echo "== $*"; ( time bin/wtzmo "$@" -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "kernel ms|real|batches in|host seconds|wall seconds" | cut -c1-260; md5sum /tmp/x.ovl | cut -c1-32; }
run --repeat 2
run --repeat 2 --pool-gb 128
run --repeat 2 --pool-gb 200
run --repeat 2 --pool-gb 200 --batch 8192
