#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
names,seqs=synth.synth_reads(2000000,20.0,seed=41,repeats=True)
print(synth.write_fasta('/tmp/rep.fa',names,seqs), len(names))
PY
A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
for kb in "20 36" "18 36" "24 36" "20 48"; do
set -- $kb
echo "== tier 3 slice $1 KB, tier 4 $2 KB"
( time env WTZ_DM_TIER3_KB=$1 WTZ_DM_TIER4_KB=$2 WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/rep.fa -fo /tmp/rp.ovl $A ) > /tmp/rp.err 2>&1
for t in 2 3 4; do grep -E "tier $t" /tmp/rp.err | awk -v t=$t '{n+=$8; s+=$(NF-1)} END{print "tier", t, "pairs", n, "ms", s}'; done
grep -E "records,|pool peak" /tmp/rp.err | cut -c1-250; md5sum /tmp/rp.ovl
done
