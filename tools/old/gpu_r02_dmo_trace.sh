#!/bin/bash
# rocprofv3 kernel statistics of the configs[2] dmo bench run only (after a dmo change)
TAG=${1:-r02q}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dmo -o dmo -- python $R/bench.py --engine dmo --no-cpu-baseline > $O/trace_dmo.log 2>&1
cd $R
python tools/summarize_profiles.py $O $O/summary
find $O -name "*kernel_trace.csv" -size +8M -delete
head -8 $O/summary/trace_dmo_kernel_stats.csv
