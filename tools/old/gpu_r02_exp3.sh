#!/bin/bash
# four-windows-per-wave K_winalign: parity (goldens + scale md5) and timing against the one-window form
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fresh or repeat" 2>&1 | tail -5
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29)); print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
run(){ tag=$1; fa=$2; shift; shift; echo "== $tag"; ( time env $ENVX bin/wtzmo -i $fa -fo /tmp/y.ovl "$@" ) > $O/$tag.err 2>&1; grep -E "records,|kernel ms|real|winalign-profile" $O/$tag.err | sort | uniq -c | sort -rn | head -8 | cut -c1-250; md5sum /tmp/y.ovl | cut -c1-12; }
Z="-k 16 -s 200 -m 0.6"
ENVX="WTZ_PROFILE_PAIR=1" run e_zmo_prof /tmp/ecoli.fa $Z
ENVX="A=1" run e_zmo /tmp/ecoli.fa $Z
ENVX="WTZ_WINALIGN4=0" run e_zmo_old /tmp/ecoli.fa $Z
ENVX="A=1" run y_zmo /tmp/yeast100.fa $Z
ENVX="WTZ_WINALIGN4=0" run y_zmo_old /tmp/yeast100.fa $Z
echo "expect ecoli 3c46e34fd78e yeast100 d532b3cbc68b"
