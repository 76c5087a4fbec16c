#!/bin/bash
set -x
mkdir -p gpurun_out
G=tests/golden
( WTZ_SW_CHECK=1 timeout 600 bin/wtzmo -i $G/tiny.fa.gz -fo /tmp/t.ovl -k 16 -s 200 -m 0.6 ) > gpurun_out/wave_check.log 2>&1
tail -4 gpurun_out/wave_check.log
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
n,s=synth.synth_reads(300000,20,seed=1)
print(synth.write_fasta('/tmp/smoke.fa',n,s), len(n))
PY
( WTZ_SW_CHECK=1 timeout 900 bin/wtzmo -i /tmp/smoke.fa -fo /tmp/smoke.chk.ovl -k 16 -s 200 -m 0.6 ) > gpurun_out/wave_check2.log 2>&1
tail -4 gpurun_out/wave_check2.log
( time timeout 900 bin/wtzmo -i /tmp/smoke.fa -fo /tmp/smoke.gpu.ovl -k 16 -s 200 -m 0.6 --stats gpurun_out/smoke.stats ) > gpurun_out/smoke_run.log 2>&1
tail -6 gpurun_out/smoke_run.log
timeout 300 oracle/_ref/wtzmo_ref -t 1 -i /tmp/smoke.fa -fo /tmp/smoke.ref.ovl -k 16 -s 200 -m 0.6 > /dev/null 2>&1
cmp /tmp/smoke.gpu.ovl /tmp/smoke.ref.ovl && echo "SMOKE600 GPU==REF" | tee -a gpurun_out/smoke_run.log
