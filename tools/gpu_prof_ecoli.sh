#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_ecoli.log 2>&1
grep -E "kernel ms|records,|speculation|host seconds" gpurun_out/bench_ecoli.log
md5sum /tmp/wtz_bench/bench_r0.ovl > gpurun_out/ecoli_ovl.md5
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ecoli -o ecoli -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_ecoli.log 2>&1
cd $GRAFT_REPO_ROOT
FA=$(ls /tmp/wtz_bench/reads_G4600000*.fa)
( time timeout 1200 oracle/_ref/wtzmo_ref -t 1 -i $FA -fo /tmp/ref1.ovl -k 16 -s 200 -m 0.6 ) > gpurun_out/ref_t1.log 2>&1
md5sum /tmp/ref1.ovl >> gpurun_out/ecoli_ovl.md5
tail -4 gpurun_out/ref_t1.log
( time timeout 1200 oracle/_ref/wtzmo_ref -t $(nproc) -i $FA -fo /tmp/refN.ovl -9 /tmp/refN.pairs -k 16 -s 200 -m 0.6 ) > gpurun_out/ref_tN.log 2>&1
tail -4 gpurun_out/ref_tN.log
wc -l /tmp/refN.pairs /tmp/refN.ovl >> gpurun_out/ref_tN.log
( time timeout 1200 oracle/_ref/wtzmo_ref -t 32 -i $FA -fo /tmp/ref32.ovl -9 /tmp/ref32.pairs -k 16 -s 200 -m 0.6 ) > gpurun_out/ref_t32.log 2>&1
tail -4 gpurun_out/ref_t32.log
