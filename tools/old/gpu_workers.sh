#!/bin/bash
export TMPDIR=/tmp
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11)
PY
bin/wtzmo -i /dev/null -fo /tmp/x 2>/dev/null
for w in 1 2 3; do
echo "== workers $w"
bin/wtzmo --workers $w --pool-gb 40 --repeat 2 -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "records,|speculation" | tail -2; md5sum /tmp/e.ovl | cut -c1-32
bin/wtzmo --workers $w --pool-gb 20 --repeat 2 -i /tmp/ecoli.fa -fo /tmp/d.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 2>&1 | grep -E "records,|speculation" | tail -2; md5sum /tmp/d.ovl | cut -c1-32
done
