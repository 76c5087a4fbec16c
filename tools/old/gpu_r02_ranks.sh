#!/bin/bash
# two ranks on this box's one GPU (gloo exchange, WTZ_BENCH_BACKEND=gloo): bench.py's N > 1 path end to end, the output of rank 0 must be the -t 1 file
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp WTZ_BENCH_TMP=/tmp/wtz_bench
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "multi_device" 2>&1 | tail -3
python bench.py --workload ecoli --no-cpu-baseline --steps 2 --warmup 1 > $O/n1.json 2> $O/n1.err; tail -1 $O/n1.json | cut -c1-200
WTZ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --workload ecoli --steps 2 --warmup 1 --pool-gb 40 > $O/n2.json 2> $O/n2.err
tail -1 $O/n2.json | cut -c1-1200; grep -E "records,|Error|error" $O/n2.err | tail -4
md5sum /tmp/wtz_bench/bench_r0.ovl /tmp/wtz_bench/bench_r1.ovl; echo "expect 3c46e34fd78ef9667fd72ad151100b59 for rank 0, empty for rank 1"
echo "== --gpu-list 0,0 on the E. coli shape"
( time bin/wtzmo --gpu-list 0,0 --pool-gb 40 -i /tmp/wtz_bench/reads_G4600000_c25_s11.fa -fo /tmp/g2.ovl -k 16 -s 200 -m 0.6 ) 2>&1 | grep -E "records,|real|kernel ms" | cut -c1-200; md5sum /tmp/g2.ovl
