#!/bin/bash
# BASELINE configs[2] (1.2 Gbp of reads) on one GPU with the DEFAULT pool: timing lines, md5 of the outputs (compared with tests/golden/big_manifest.json when it has them)
TAG=${1:-r02b}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
t=time.time(); print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29), 'gen %.1fs' % (time.time()-t))
PY
for e in zmo dmo; do
  if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
  echo "== $e"
  ( time timeout 1200 bin/wtzmo $WTZ_EXTRA -i /tmp/yeast100.fa -fo /tmp/y100.$e.ovl --stats $O/y100.$e.stats $A ) 2> $O/y100.$e.err
  grep -E "records,|kernel ms|batches|real|error|scratch|split|host seconds|wall seconds" $O/y100.$e.err
  md5sum /tmp/y100.$e.ovl /tmp/y100.$e.ovl.contained | tee $O/y100.$e.md5
  wc -l /tmp/y100.$e.ovl
done
