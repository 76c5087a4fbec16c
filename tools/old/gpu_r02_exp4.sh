#!/bin/bash
# timing + md5 of both workloads / both engines (quick regression line per change)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r02i}; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29)); print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
run(){ tag=$1; fa=$2; shift; shift; echo "== $tag"; ( time env $ENVX bin/wtzmo -i $fa -fo /tmp/y.ovl "$@" ) > $O/$tag.err 2>&1; grep -E "records,|kernel ms|real" $O/$tag.err | cut -c1-250; md5sum /tmp/y.ovl | cut -c1-12; }
Z="-k 16 -s 200 -m 0.6"; D="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
ENVX="A=1"
run e_zmo /tmp/ecoli.fa $Z
run e_dmo /tmp/ecoli.fa $D
run y_zmo /tmp/yeast100.fa $Z
[ -z "$SKIP_YDMO" ] && run y_dmo /tmp/yeast100.fa $D
echo "expect ecoli zmo 3c46e34fd78e dmo aaeb67d219a9 ; yeast100 zmo d532b3cbc68b dmo fda714356c78"
