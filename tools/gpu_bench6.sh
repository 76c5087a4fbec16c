#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
for w in 1 2 3; do
  echo "== workers $w"
  bin/wtzmo --pool-gb 60 --workers $w -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "records,|batches|kernel ms|host seconds|split"
  md5sum /tmp/e.ovl
done > gpurun_out/workers.log 2>&1
cat gpurun_out/workers.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
