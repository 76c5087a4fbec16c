#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
for A in "-k 16 -s 200 -m 0.6" "-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; do
echo "== $A"; ( time bin/wtzmo --repeat 4 -i /tmp/yeast100.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "records,|kernel ms|real" | cut -c1-120; md5sum /tmp/x.ovl | cut -c1-32
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden" 2>&1 | tail -2
