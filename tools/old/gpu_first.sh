#!/bin/bash
# first contact with the GPU: smoke, a timed run with per-stage kernel times, then the parity tests
set -x
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > gpurun_out/rocminfo.txt 2>&1
( time timeout 600 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1
tail -5 gpurun_out/smoke.log
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
n,s=synth.synth_reads(300000,20,seed=1)
print(synth.write_fasta('/tmp/smoke.fa',n,s), len(n))
PY
( time timeout 900 bin/wtzmo -i /tmp/smoke.fa -fo /tmp/smoke.gpu.ovl -k 16 -s 200 -m 0.6 --stats gpurun_out/smoke.stats ) > gpurun_out/smoke_run.log 2>&1
tail -8 gpurun_out/smoke_run.log
( time timeout 300 oracle/_ref/wtzmo_ref -t 1 -i /tmp/smoke.fa -fo /tmp/smoke.ref.ovl -k 16 -s 200 -m 0.6 ) > gpurun_out/smoke_ref.log 2>&1
cmp /tmp/smoke.gpu.ovl /tmp/smoke.ref.ovl && echo "SMOKE600 GPU==REF" | tee -a gpurun_out/smoke_run.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
