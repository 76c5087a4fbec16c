#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ f=$1; shift; echo "== $f: $*"; ( time env "$@" bin/wtzmo -i /tmp/$f.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "kernel ms" | sed 's/.*stitch/stitch/' | cut -c1-70; md5sum /tmp/x.ovl | cut -c1-32; }
for f in ecoli yeast100; do
run $f WTZ_X=0
run $f WTZ_SW_MW_TOP=64
run $f WTZ_SW_MW_MIN=2048
run $f WTZ_REG_SPLIT=1
run $f WTZ_REG_SPLIT=1 WTZ_SW_MW_TOP=64
run $f WTZ_REG_SPLIT=1 WTZ_SW_MW_TOP=256
run $f WTZ_REG_SPLIT=1 WTZ_SW_MW_MIN=2048
run $f WTZ_REG_SPLIT=2 WTZ_SW_MW_TOP=64
run $f WTZ_REG_SPLIT=2 WTZ_SW_MW_MIN=2048
run $f WTZ_X=0
done
