#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
n,s=synth.synth_reads(300000,20,seed=1)
print(synth.write_fasta('/tmp/smoke.fa',n,s), len(n))
PY
( WTZ_SW_CHECK=1 timeout 900 bin/wtzmo -i /tmp/smoke.fa -fo /tmp/smoke.chk.ovl -k 16 -s 200 -m 0.6 ) > gpurun_out/wave_check2.log 2>&1
tail -3 gpurun_out/wave_check2.log
( time timeout 2400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/bench_ecoli.log 2>&1
grep -E "kernel ms|records,|speculation|host seconds" gpurun_out/bench_ecoli.log
md5sum /tmp/wtz_bench/bench_r0.ovl > gpurun_out/ecoli_ovl.md5
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
