#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
echo "== zmo w1"; bin/wtzmo --pool-gb 60 --workers 1 -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "records,|batches|kernel ms|host seconds|split"; md5sum /tmp/e.ovl
for w in 1 2 3; do echo "== dmo w$w"; bin/wtzmo --pool-gb 40 --workers $w -i /tmp/ecoli.fa -fo /tmp/d.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 2>&1 | grep -E "records,|batches|kernel ms|host seconds|split"; md5sum /tmp/d.ovl; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
