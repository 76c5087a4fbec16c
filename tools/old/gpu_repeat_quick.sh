#!/bin/bash
# product only on the repeat-rich 2 Mbp x20 set; reference -t 1 md5s (tools/gpu_repeat_shape.sh): zmo 2b39d5800eebd954ea359aaffa1a3b86 dmo 0c0c7770300c6077e66ade7354e9ce41
export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
names,seqs=synth.synth_reads(2000000,20,seed=77,repeats=True)
synth.write_fasta('/tmp/rep2m.fa',names,seqs)
PY
for pg in ${WTZ_POOLS:-100 150}; do for e in ${WTZ_ENGINES:-zmo dmo}; do
  if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
  echo "== $e product, pool $pg GB"; ( time timeout 900 bin/wtzmo --pool-gb $pg -i /tmp/rep2m.fa -fo /tmp/rp.$e.ovl $A ) 2>&1 | grep -E "records,|kernel ms|real|fault" | cut -c1-220 | tail -4; md5sum /tmp/rp.$e.ovl
done; done
