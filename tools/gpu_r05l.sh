#!/bin/bash
# round 5: the K-sw3 stage of a range is bound by its LONGEST job (5 800 jobs of one call: 6.7 ms for 1.5 ms of rows), and so are the other per-range stages:
# several contexts on the one device run the parts of a range side by side - does the device fill?  one / two / three / four contexts, whole step, md5
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05l}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_gen.json 2> $O/bench_gen.err
FA=/tmp/wtz_bench/reads_G12000000_c100_s29.fa
for L in 0 0,0 0,0,0 0,0,0,0; do
  N=$(echo $L | tr ',' '\n' | wc -l)
  ( time timeout 900 bin/wtzmo -i $FA -fo /tmp/wtz_bench/m.ovl --gpu-list $L --pool-gb $((96 / N)) --repeat 3 -k 16 -s 200 -m 0.6 ) > $O/ctx$N.out 2> $O/ctx$N.err
  echo "== contexts $N"; grep "records,\|kernel ms" $O/ctx$N.err | tail -4; md5sum /tmp/wtz_bench/m.ovl | cut -c1-32; rm -f /tmp/wtz_bench/m.ovl*
done
for L in 0 0,0; do
  N=$(echo $L | tr ',' '\n' | wc -l)
  ( time timeout 900 bin/wtzmo -i $FA -fo /tmp/wtz_bench/m.ovl --gpu-list $L --pool-gb $((96 / N)) --repeat 2 -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 ) > $O/dmo_ctx$N.out 2> $O/dmo_ctx$N.err
  echo "== dmo contexts $N"; grep "records,\|kernel ms" $O/dmo_ctx$N.err | tail -2; md5sum /tmp/wtz_bench/m.ovl | cut -c1-32; rm -f /tmp/wtz_bench/m.ovl*
done
