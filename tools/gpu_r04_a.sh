#!/bin/bash
# round 4, first GPU run: whole GPU suite on the new host driver (pairs dealt by candidate, split z-index, writer-thread formatting, binary hand-off),
# the 2-rank bench form on the one-GPU box (gloo stand-in), the default bench line's host timers, and the two-context model run for the N-GPU budget
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04a}; mkdir -p $O
export TMPDIR=/tmp WTZ_TEST_KEEP_STDERR=$O/stderr
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time WTZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload ecoli --no-cpu-baseline ) > $O/bench_ecoli_2ranks_gloo.json 2> $O/bench_ecoli_2ranks_gloo.err; tail -1 $O/bench_ecoli_2ranks_gloo.json | cut -c1-400; grep -c . $O/bench_ecoli_2ranks_gloo.err
( time timeout 1200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 ) > $O/bench_zmo.json 2> $O/bench_zmo.err; tail -1 $O/bench_zmo.json | cut -c1-300; grep "host seconds\|commit sections\|records," $O/bench_zmo.err | tail -4
FA=$(ls /tmp/wtz_bench/reads_G12000000_c100_s29.fa)
( time timeout 900 bin/wtzmo -i $FA -fo /tmp/wtz_bench/m2.ovl --gpu-list 0,0 --pool-gb 48 --repeat 2 -k 16 -s 200 -m 0.6 ) > $O/model_2ctx.out 2> $O/model_2ctx.err; grep "host seconds\|commit sections\|records,\|kernel ms" $O/model_2ctx.err | tail -4; md5sum /tmp/wtz_bench/m2.ovl
( time WTZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline --steps 2 --warmup 1 --pool-gb 48 ) > $O/bench_yeast_2ranks_gloo.json 2> $O/bench_yeast_2ranks_gloo.err; tail -1 $O/bench_yeast_2ranks_gloo.json | cut -c1-300; grep "host seconds\|commit sections" $O/bench_yeast_2ranks_gloo.err | tail -2
