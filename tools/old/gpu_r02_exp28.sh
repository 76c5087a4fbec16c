#!/bin/bash
# seed lookup of the next batch beside the pair stages (candidate pool + side stream) against the synchronous form (WTZ_CAND_SYNC=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
run(){ f=$1; A=$2; shift 2; echo "== $f [$A]: $*"; ( time env "$@" bin/wtzmo --repeat 2 -i /tmp/$f.fa -fo /tmp/x.ovl $A ) 2>&1 | grep -E "records,|kernel ms|cand-profile|batches in|error|failed" | tail -5 | cut -c1-230; md5sum /tmp/x.ovl | cut -c1-32; }
Z="-k 16 -s 200 -m 0.6"; D="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
run ecoli "$Z" WTZ_PROFILE_PAIR=0
run ecoli "$Z" WTZ_CAND_SYNC=1
run yeast100 "$Z" WTZ_X=1
run yeast100 "$Z" WTZ_CAND_SYNC=1
run yeast100 "$D" WTZ_X=1
run yeast100 "$D" WTZ_CAND_SYNC=1
