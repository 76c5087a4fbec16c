#!/bin/bash
# repeat-rich synthetic input (tandem arrays + dispersed 6-kb copies in a 2 Mbp genome, x20): product vs reference -t 32 wall time, md5 vs -t 1
export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
names,seqs=synth.synth_reads(2000000,20,seed=77,repeats=True)
synth.write_fasta('/tmp/rep2m.fa',names,seqs); print(len(names), sum(s.size for s in seqs))
PY
for e in zmo dmo; do
  if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
  echo "== $e product"; ( time WTZ_PROFILE_PAIR=1 bin/wtzmo --pool-gb 100 -i /tmp/rep2m.fa -fo /tmp/rp.$e.ovl $A ) 2>&1 | grep -E "records,|kernel ms|real|tier" | cut -c1-220 | tail -12; md5sum /tmp/rp.$e.ovl
  echo "== $e reference -t 32"; ( time oracle/_ref/wtzmo_ref -t 32 -f -i /tmp/rep2m.fa -o /tmp/rr32.$e.ovl $A ) 2>&1 | grep real
  echo "== $e reference -t 1"; ( time timeout 1500 oracle/_ref/wtzmo_ref -t 1 -f -i /tmp/rep2m.fa -o /tmp/rr.$e.ovl $A ) 2>&1 | grep real; md5sum /tmp/rr.$e.ovl
done
