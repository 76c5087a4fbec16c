#!/bin/bash
# BASELINE configs[2] size: 1.2 Gbp of synthetic reads (12 Mbp genome x 100) on one GPU, product only: does it fit and how fast is it
export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
t=time.time(); print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29), 'gen %.1fs' % (time.time()-t))
PY
for e in zmo dmo; do
  if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
  echo "== $e"
  ( time timeout 1200 bin/wtzmo --pool-gb 200 -i /tmp/yeast100.fa -fo /tmp/y100.$e.ovl $A ) 2>&1 | grep -E "records,|kernel ms|batches|real|error|scratch"
  ls -la /tmp/y100.$e.ovl | awk '{print $5}'
done
