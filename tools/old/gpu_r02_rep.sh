#!/bin/bash
# repeat-rich set (big manifest "repeat": 2 Mbp genome, x20, seed 41): dmo tiers and phase split
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
names,seqs=synth.synth_reads(2000000,20.0,seed=41,repeats=True)
print(synth.write_fasta('/tmp/rep.fa',names,seqs), len(names))
PY
A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
( time env WTZ_PROFILE_PAIR=1 bin/wtzmo $WTZ_EXTRA -i /tmp/rep.fa -fo /tmp/rp.ovl $A ) > /tmp/rp.err 2>&1
grep -E "tier|K_pair first" /tmp/rp.err | cut -c1-200
grep -E "pair-profile\] n=" /tmp/rp.err | awk '{split($8,a,"/"); m+=a[1]; s+=a[2]; w+=a[3]; t+=a[4]} END{print "kticks match/sort/denoise/total", m, s, w, t}'
grep -E "records,|kernel ms|real|batches in" /tmp/rp.err | cut -c1-250; md5sum /tmp/rp.ovl
