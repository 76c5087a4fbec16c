#!/bin/bash
# -n (kswx_refine_alignment) at E. coli shape: product vs reference -t 1, md5 of the full .ovl
export TMPDIR=/tmp
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11)
PY
( time bin/wtzmo --pool-gb 100 -i /tmp/ecoli.fa -fo /tmp/n.ovl -k 16 -s 200 -m 0.6 -n ) 2>&1 | grep -E "records,|real"
md5sum /tmp/n.ovl
( time timeout 2400 oracle/_ref/wtzmo_ref -t 1 -f -i /tmp/ecoli.fa -o /tmp/rn.ovl -k 16 -s 200 -m 0.6 -n ) 2>&1 | grep real
md5sum /tmp/rn.ovl
