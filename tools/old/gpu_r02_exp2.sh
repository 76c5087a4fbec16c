#!/bin/bash
# candidate context experiment: yeast100 + ecoli, both engines, with / without the clone
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29)); print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
run(){ tag=$1; fa=$2; shift; shift; echo "== $tag"; ( time bin/wtzmo -i $fa -fo /tmp/y.ovl "$@" ) > $O/$tag.err 2>&1; grep -E "records,|kernel ms|real|wall seconds|host seconds|split" $O/$tag.err | cut -c1-260; md5sum /tmp/y.ovl | cut -c1-12; }
Z="-k 16 -s 200 -m 0.6"; D="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
run y_zmo /tmp/yeast100.fa $Z
run y_zmo_nocc /tmp/yeast100.fa $Z --no-cand-ctx
run y_dmo /tmp/yeast100.fa $D
run e_zmo /tmp/ecoli.fa $Z
run e_zmo_nocc /tmp/ecoli.fa $Z --no-cand-ctx
run e_dmo /tmp/ecoli.fa $D
