#!/bin/bash
# env-switch A/B on both workloads (zmo): each line = one run's kernel ms + md5
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ tag=$1; f=$2; shift 2; echo "== $tag $f: $*"; ( time env "$@" bin/wtzmo -i /tmp/$f.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "kernel ms|real|winalign-profile" | cut -c1-200; md5sum /tmp/x.ovl | cut -c1-32; }
run base ecoli WTZ_X=0
run wa4 ecoli WTZ_WINALIGN4=1
run base yeast100 WTZ_X=0
run wa4 yeast100 WTZ_WINALIGN4=1
