#!/bin/bash
# one-engine refresh of the profile summaries on the current kernel sources (the tail of tools/the round-4 final script cut to what fits a few GPU-minutes):
# rocprofv3 kernel statistics + the three PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_* - own runs, no trace domains) of `bench.py --steps 1`.
# usage: tools/gpu_r05_refresh.sh <tag> [zmo|dmo]    -> gpurun_out/<tag>/summary/ (copied to profiles/ by hand, with the .meta.json sidecar bench.py reads)
TAG=${1:-r05r}
E=${2:-zmo}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$E -o $E -- python $R/bench.py --engine $E --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/trace_$E.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o $E -- python $R/bench.py --engine $E --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_$c.log 2>&1
done
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o $E -- python $R/bench.py --engine $E --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_SQ.log 2>&1
cd $R
mkdir -p $O/split_$E $O/summary
for d in trace_$E pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do [ -d $O/$d ] && ln -s $O/$d $O/split_$E/$d; done
python tools/summarize_profiles.py $O/split_$E $O/summary_$E
KID=$(python3 -c "import bench; print(bench.kernel_source_id())")
[ -f $O/summary_$E/pmc_per_kernel.csv ] && { cp $O/summary_$E/pmc_per_kernel.csv $O/summary/r05_yeast100_${E}_pmc_per_kernel.csv; echo "{\"kernel_source_id\": \"$KID\", \"command\": \"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python bench.py --engine $E --steps 1 --warmup 0\"}" > $O/summary/r05_yeast100_${E}_pmc_per_kernel.csv.meta.json; }
[ -f $O/summary_$E/trace_${E}_kernel_stats.csv ] && cp $O/summary_$E/trace_${E}_kernel_stats.csv $O/summary/r05_yeast100_${E}_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
rm -rf $O/split_$E
ls $O/summary
