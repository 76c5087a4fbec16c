#!/bin/bash
# Round-3: what does NOT divide by N.  configs[2] zmo with 1 / 2 / 4 contexts on the ONE device of the box (--gpu-list 0,0,...: same dealing,
# same central commit as --gpus N; the device work does not get faster, the host side shows what rank 0 / the commit thread adds per part),
# two torchrun ranks over gloo on the same device (the exchange path), two worker contexts, and the K-sw3 band-class split.
TAG=${1:-r03model}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
for cfg in WTZ_EXT_SPLIT=0 WTZ_EXT_SPLIT=1; do
  env $cfg python bench.py --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  echo $cfg; tail -1 $O/bench_$cfg.json | cut -c1-180; grep "kernel ms" $O/bench_$cfg.err | tail -1
done
FA=$(ls /tmp/wtz_bench/reads_G12000000_c100_s29.fa)
n=1
for cfg in "--gpu-list 0 --pool-gb 120" "--gpu-list 0,0 --pool-gb 60" "--gpu-list 0,0,0,0 --pool-gb 30" "--workers 2 --pool-gb 60" "--workers 2 --pool-gb 60 --batch 1024"; do
  timeout 600 bin/wtzmo -i $FA -fo /tmp/wtz_bench/model.ovl -k 16 -s 200 -m 0.6 --repeat 3 $cfg 2> $O/model_$n.err
  echo "== $cfg"; md5sum /tmp/wtz_bench/model.ovl | cut -c1-32; grep -E "records|host seconds|wall seconds|batches in|kernel ms" $O/model_$n.err | tail -5
  n=$((n+1))
done
rm -f /tmp/wtz_bench/model.ovl*
WTZ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --pool-gb 60 > $O/bench_ranks2.json 2> $O/bench_ranks2.err
tail -1 $O/bench_ranks2.json | cut -c1-200; grep -E "host seconds|batches in" $O/bench_ranks2.err | tail -2
