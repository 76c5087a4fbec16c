#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/bench_ecoli.log 2>&1
grep -E "kernel ms|records,|speculation|host seconds" gpurun_out/bench_ecoli.log
md5sum /tmp/wtz_bench/bench_r0.ovl > gpurun_out/ecoli_ovl.md5
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
