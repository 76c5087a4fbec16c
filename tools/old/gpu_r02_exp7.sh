#!/bin/bash
# function-level DP parity (every device form) + golden parity subset, then timing on both workloads (zmo; dmo on ecoli)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dp_forms.py -q -x 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fresh" 2>&1 | tail -3
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
A="-k 16 -s 200 -m 0.6"
run(){ tag=$1; f=$2; shift 2; echo "== $tag $f: $*"; ( time env "$@" bin/wtzmo -i /tmp/$f.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "kernel ms|real" | cut -c1-200; md5sum /tmp/x.ovl | cut -c1-32; }
run base ecoli WTZ_X=0
run base yeast100 WTZ_X=0
[ -n "$EXTRA_ENV" ] && { run extra ecoli $EXTRA_ENV; run extra yeast100 $EXTRA_ENV; }
