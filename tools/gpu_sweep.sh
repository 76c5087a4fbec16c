#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
for cfg in "64 2048" "256 2048" "256 4096" "1024 4096" "4096 4096"; do
  set -- $cfg
  echo "== first $1 max $2"
  bin/wtzmo --pool-gb 120 --first-batch $1 --batch $2 -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "records,|speculation|kernel ms|retry"
  md5sum /tmp/e.ovl
done > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
