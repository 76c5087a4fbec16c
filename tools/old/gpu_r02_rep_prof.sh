#!/bin/bash
# repeat-rich set, dmo, phase profiler build: where do the big-tier pairs spend their time?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
names,seqs=synth.synth_reads(2000000,20.0,seed=41,repeats=True)
print(synth.write_fasta('/tmp/rep.fa',names,seqs), len(names))
PY
WTZ_HIPCC_FLAGS="-DWTZ_PROFILE" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
( time env WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/rep.fa -fo /tmp/rp.ovl $A ) > /tmp/rp.err 2>&1
grep -E "tier|K_pair first" /tmp/rp.err | cut -c1-200
grep -E "phase-profile" /tmp/rp.err | cut -c1-1200
grep -E "records,|kernel ms|real|batches in" /tmp/rp.err | cut -c1-250; md5sum /tmp/rp.ovl
