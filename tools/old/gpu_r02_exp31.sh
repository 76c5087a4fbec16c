#!/bin/bash
# dmo first-launch slice with the pool image allowed: 24 KB (default) against 20 / 32 KB (compile-time), configs[2] + repeat-rich + heavy-path tests
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
from smartdenovo_amd import synth
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
names,seqs=synth.synth_reads(2000000,20.0,seed=41,repeats=True)
print(synth.write_fasta('/tmp/rep.fa',names,seqs), len(names))
PY
D="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"

for f in "-DWTZ_PAIR_DM_LDS_BYTES=18432" "-DWTZ_PAIR_DM_LDS_BYTES=22528"; do
  WTZ_HIPCC_FLAGS="$f" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  for s in yeast100 rep; do bin/wtzmo --repeat 2 -i /tmp/$s.fa -fo /tmp/v.ovl $D 2>&1 | grep -E "records," | tail -1 | cut -c1-120; md5sum /tmp/v.ovl | cut -c1-32; done
done
