#!/bin/bash
# Round-3 final validation + measurement on the GPU box: whole GPU test-suite, the driver's exact bench command, the dmo line, smoke(), rocprofv3 kernel stats of both engines.
# (PMC passes: tools/gpu_r03_dmo_pmc.sh for dmo; for zmo the loop below with ZMO_PMC=1.)   usage: tools/gpu_r03_final.sh <tag>   -> gpurun_out/<tag>/, summaries copied to profiles/ by hand
TAG=${1:-r03final}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_zmo.json 2> $O/bench_zmo.err; tail -1 $O/bench_zmo.json | cut -c1-260; grep real $O/bench_zmo.err
python bench.py --engine dmo --steps 3 --warmup 2 > $O/bench_dmo.json 2>/dev/null; tail -1 $O/bench_dmo.json | cut -c1-200
python bench.py --workload ecoli --steps 5 --warmup 2 > $O/bench_ecoli_zmo.json 2>/dev/null
python bench.py --workload ecoli --engine dmo --steps 5 --warmup 2 > $O/bench_ecoli_dmo.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/trace_zmo.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dmo -o dmo -- python $R/bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline > $O/trace_dmo.log 2>&1
if [ -n "$ZMO_PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do timeout 1500 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1; done
  timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_SQ.log 2>&1
fi
cd $R
python tools/summarize_profiles.py $O $O/summary
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls $O/summary
