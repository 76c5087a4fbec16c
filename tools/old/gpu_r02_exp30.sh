#!/bin/bash
# dmo: oversize strands stay in the first K_pair launch (image in the pool) - timing on configs[2] / configs[1] / the repeat-rich set + parity
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
from smartdenovo_amd import synth
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
names,seqs=synth.synth_reads(2000000,20.0,seed=41,repeats=True)
print(synth.write_fasta('/tmp/rep.fa',names,seqs), len(names))
PY
D="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"
for f in ecoli yeast100 rep; do
echo "== $f"; ( time env WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/$f.fa -fo /tmp/x.ovl $D ) > /tmp/x.err 2>&1
grep -E "tier" /tmp/x.err | awk '{n[$4]+=$8; t[$4]+=$(NF-1)} END{for(k in n) print "  tier", k, "pairs", n[k], "ms", t[k]}'
grep -E "K_pair first" /tmp/x.err | awk '{n+=$5; t+=$(NF-1)} END{print "  first launch pairs", n, "ms", t}'
grep -E "records,|kernel ms" /tmp/x.err | cut -c1-150; md5sum /tmp/x.ovl | cut -c1-32
done
echo "expect ecoli dmo $(python -c "import json; m=json.load(open('tests/golden/big_manifest.json')); print(m['cases']['ecoli_dmo']['md5_full'], m['cases']['yeast100_dmo']['md5_full'], m['cases']['repeat_dmo']['md5_full'])")"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fresh" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_scale.py -q -x -k "heavy" 2>&1 | tail -2
