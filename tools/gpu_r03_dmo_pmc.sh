#!/bin/bash
# PMC passes of one configs[2] dmo step (FETCH_SIZE / WRITE_SIZE in separate runs, then the SQ set): the `traffic` of the dmo bench line's roofline_zmer
TAG=${1:-r03dmopmc}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1500 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o dmo -- python $R/bench.py --engine dmo --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o dmo -- python $R/bench.py --engine dmo --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_SQ.log 2>&1
cd $R
python tools/summarize_profiles.py $O $O/summary
find $O -name "*counter_collection.csv" -size +8M -delete
head -6 $O/summary/pmc_per_kernel.csv | cut -c1-200
