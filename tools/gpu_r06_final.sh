#!/bin/bash
# Round-6 measurement on the GPU box (tools/gpu_r04_final.sh with this round's names) (every step under its own timeout): the whole GPU suite, the driver's exact bench command, the dmo / E. coli-shape / 2-rank /
# configs[3]-shape lines, rocprofv3 kernel statistics and the PMC passes (SQ_* / FETCH_SIZE / WRITE_SIZE only: a pass with TA_* / TCP_* counters hung a box in this round).
# usage: tools/gpu_r06_final.sh <tag>     outputs under gpurun_out/<tag>/; summary/ holds what is copied to profiles/ (with the .meta.json sidecars bench.py reads)
TAG=${1:-r06final}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp WTZ_TEST_KEEP_FLY=1 WTZ_TEST_KEEP_STDERR=$O/stderr
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-28s n_gpus %d  %.3f s/step  %.2f Gbp/s  parity %s  roofline.frac %.4f  cpu %s | %s" % (sys.argv[2], d['n_gpus'], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('cpu_baseline') or {}).get('value'), {a:round(b) for a,b in k.items()}))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; tail -24 $O/pytest.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; line $O/bench_driver_cmd.json "driver cmd (zmo, configs[2])"; grep real $O/bench_driver_cmd.err
( time timeout 1200 python bench.py --engine dmo ) > $O/bench_dmo.json 2> $O/bench_dmo.err; line $O/bench_dmo.json "dmo configs[2]"
timeout 600 python bench.py --workload ecoli > $O/bench_ecoli_zmo.json 2> $O/bench_ecoli_zmo.err; line $O/bench_ecoli_zmo.json "zmo configs[1]"
timeout 600 python bench.py --workload ecoli --engine dmo --no-cpu-baseline > $O/bench_ecoli_dmo.json 2> $O/bench_ecoli_dmo.err; line $O/bench_ecoli_dmo.json "dmo configs[1]"
WTZ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload ecoli --no-cpu-baseline > $O/bench_ecoli_2ranks_gloo.json 2> $O/bench_ecoli_2ranks_gloo.err; line $O/bench_ecoli_2ranks_gloo.json "2 ranks on one GPU (gloo)"
WTZ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 2 --warmup 1 --pool-gb 48 > $O/bench_yeast_2ranks_gloo.json 2> $O/bench_yeast_2ranks_gloo.err; line $O/bench_yeast_2ranks_gloo.json "2 ranks, configs[2] (gloo)"
FA=/tmp/wtz_bench/reads_G12000000_c100_s29.fa
( time timeout 600 bin/wtzmo -i $FA -fo /tmp/wtz_bench/m2.ovl --gpu-list 0,0 --pool-gb 48 --repeat 2 -k 16 -s 200 -m 0.6 ) > $O/model_2ctx.out 2> $O/model_2ctx.err; grep "host seconds\|commit sections\|records," $O/model_2ctx.err | tail -3; md5sum /tmp/wtz_bench/m2.ovl; rm -f /tmp/wtz_bench/m2.ovl*
( time timeout 600 bin/wtzmo -i $FA -fo /tmp/wtz_bench/b.ovlb --binary-out -k 16 -s 200 -m 0.6 ) > $O/binary_out.out 2> $O/binary_out.err; grep "records,\|host seconds" $O/binary_out.err | tail -2; ls -la /tmp/wtz_bench/b.ovlb | awk '{print $5}'; rm -f /tmp/wtz_bench/b.ovlb*
( time timeout 1800 python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_fly70.json 2> $O/bench_fly70.err; line $O/bench_fly70.json "zmo configs[3] shape, one GPU"; grep "records,\|host seconds\|kernel ms\|z-mer index" $O/bench_fly70.err | tail -4; rm -f /tmp/wtz_bench/reads_G140000000_*
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 > $O/trace_zmo.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dmo -o dmo -- python $R/bench.py --engine dmo --no-cpu-baseline --no-verify --steps 1 --warmup 1 > $O/trace_dmo.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/dmo_pmc_$c -o dmo -- python $R/bench.py --engine dmo --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/dmo_pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_SQ.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/dmo_pmc_SQ -o dmo -- python $R/bench.py --engine dmo --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/dmo_pmc_SQ.log 2>&1
cd $R
mkdir -p $O/split_zmo $O/split_dmo
for d in trace_zmo pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do [ -d $O/$d ] && ln -s $O/$d $O/split_zmo/$d; done
[ -d $O/trace_dmo ] && ln -s $O/trace_dmo $O/split_dmo/trace_dmo
for c in FETCH_SIZE WRITE_SIZE SQ; do [ -d $O/dmo_pmc_$c ] && ln -s $O/dmo_pmc_$c $O/split_dmo/pmc_$c; done
python tools/summarize_profiles.py $O/split_zmo $O/summary_zmo; python tools/summarize_profiles.py $O/split_dmo $O/summary_dmo
mkdir -p $O/summary
KID=$(python3 -c "import bench; print(bench.kernel_source_id())")
for e in zmo dmo; do
  [ -f $O/summary_$e/pmc_per_kernel.csv ] && { cp $O/summary_$e/pmc_per_kernel.csv $O/summary/r06_yeast100_${e}_pmc_per_kernel.csv; echo "{\"kernel_source_id\": \"$KID\", \"command\": \"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python bench.py --engine $e --steps 1 --warmup 0\"}" > $O/summary/r06_yeast100_${e}_pmc_per_kernel.csv.meta.json; }
  [ -f $O/summary_$e/trace_${e}_kernel_stats.csv ] && cp $O/summary_$e/trace_${e}_kernel_stats.csv $O/summary/r06_yeast100_${e}_kernel_stats.csv
done
python3 tools/analysis/idle_gaps.py $O/trace_zmo > $O/summary/r06_device_idle_gaps_zmo.txt 2>&1; python3 tools/analysis/idle_gaps.py $O/trace_dmo > $O/summary/r06_device_idle_gaps_dmo.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
rm -rf $O/split_zmo $O/split_dmo
ls $O/summary
