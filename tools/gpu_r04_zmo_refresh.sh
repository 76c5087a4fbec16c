#!/bin/bash
# zmo-only refresh of the profile summaries on the current kernel sources (the tail of tools/gpu_r04_final.sh cut to what fits a few GPU-minutes):
# rocprofv3 kernel statistics + the three PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_* - own runs, no trace domains) of `bench.py --steps 1`.
# usage: tools/gpu_r04_zmo_refresh.sh <tag>    -> gpurun_out/<tag>/summary/ (copied to profiles/ by hand, with the .meta.json sidecar bench.py reads)
TAG=${1:-r04zr}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/trace_zmo.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_$c.log 2>&1
done
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_SQ.log 2>&1
cd $R
mkdir -p $O/split_zmo $O/summary
for d in trace_zmo pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do [ -d $O/$d ] && ln -s $O/$d $O/split_zmo/$d; done
python tools/summarize_profiles.py $O/split_zmo $O/summary_zmo
KID=$(python3 -c "import bench; print(bench.kernel_source_id())")
[ -f $O/summary_zmo/pmc_per_kernel.csv ] && { cp $O/summary_zmo/pmc_per_kernel.csv $O/summary/r04_yeast100_zmo_pmc_per_kernel.csv; echo "{\"kernel_source_id\": \"$KID\", \"command\": \"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python bench.py --engine zmo --steps 1 --warmup 0\"}" > $O/summary/r04_yeast100_zmo_pmc_per_kernel.csv.meta.json; }
[ -f $O/summary_zmo/trace_zmo_kernel_stats.csv ] && cp $O/summary_zmo/trace_zmo_kernel_stats.csv $O/summary/r04_yeast100_zmo_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
rm -rf $O/split_zmo
ls $O/summary
