#!/bin/bash
# step wall time (the "records," line: whole overlap phase) for pool sizes and for two contexts on one device
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ echo "== $*"; ( time bin/wtzmo "$@" -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "records,|kernel ms|real" | cut -c1-200; md5sum /tmp/x.ovl | cut -c1-32; }
run --repeat 3
run --repeat 3 --pool-gb 96
run --repeat 3 --pool-gb 128
run --repeat 3 --gpu-list 0,0 --pool-gb 64
run --repeat 3 --gpu-list 0,0 --pool-gb 32
