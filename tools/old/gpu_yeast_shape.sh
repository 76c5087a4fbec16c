#!/bin/bash
# larger parity check: synthetic 12 Mbp genome x 30 (yeast-shape genome size, 360 Mbp of reads): product vs reference -t 1, both engines
export TMPDIR=/tmp
mkdir -p gpurun_out
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
t=time.time(); print(bench.gen_reads('/tmp/yeast.fa',12000000,30.0,23), 'gen %.1fs' % (time.time()-t))
PY
for e in zmo dmo; do
  if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
  echo "== $e product"
  ( time bin/wtzmo --pool-gb 120 -i /tmp/yeast.fa -fo /tmp/y.$e.ovl $A ) 2>&1 | grep -E "records,|kernel ms|batches|split|real"
  md5sum /tmp/y.$e.ovl
done
[ -n "$WTZ_YEAST_PRODUCT_ONLY" ] && { echo "reference -t 1 md5s recorded in profiles/r01_yeast_shape_parity.txt: zmo 07beaa277bbaa34d43e326a0fb7bc8d1 dmo 397958335784018f8f4ed61773aceafb"; exit 0; }
echo "== dmo reference -t 1"; ( time timeout 2400 oracle/_ref/wtzmo_ref -t 1 -f -i /tmp/yeast.fa -o /tmp/r.dmo.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 ) 2>&1 | grep real; md5sum /tmp/r.dmo.ovl
echo "== zmo reference -t 1"; ( time timeout 2400 oracle/_ref/wtzmo_ref -t 1 -f -i /tmp/yeast.fa -o /tmp/r.zmo.ovl -k 16 -s 200 -m 0.6 ) 2>&1 | grep real; md5sum /tmp/r.zmo.ovl
echo "== zmo reference -t 32"; ( time timeout 1200 oracle/_ref/wtzmo_ref -t 32 -f -i /tmp/yeast.fa -o /tmp/r32.zmo.ovl -k 16 -s 200 -m 0.6 ) 2>&1 | grep real
