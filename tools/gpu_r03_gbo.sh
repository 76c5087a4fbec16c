#!/bin/bash
# f1 measurement on the GPU box: the zmo pipeline steps 2 + 3 on the configs[1]-shape input (E. coli shape, 115 Mbp):
#   bin/wtzmo | cut -f1-16 -> bin/wtgbo   (timed, rocprofv3 kernel stats of the wtgbo run)
#   reference wtgbo -t 1 (parity: same bytes) and -t 32 (the CPU baseline on this host) on the SAME overlap file
TAG=${1:-r03gbo}
GEN=${2:-4600000}; COV=${3:-25.0}; SEED=${4:-11}; REF1=${5:-1}        # default = configs[1] shape; "12000000 100.0 29 0" = configs[2] shape without the (long) reference -t 1 run
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
mkdir -p /tmp/wtz_bench
FA=/tmp/wtz_bench/gbo_G${GEN}.fa
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
print(bench.gen_reads("$FA", $GEN, $COV, $SEED))
PY
( time bin/wtzmo -i $FA -fo - -k 16 -s 200 -m 0.6 2> $O/wtzmo.err | cut -f1-16 > /tmp/wtz_bench/gbo_ecoli.ovl16 ) 2> $O/wtzmo.time; wc -l /tmp/wtz_bench/gbo_ecoli.ovl16; grep real $O/wtzmo.time
( time bin/wtgbo -i $FA -j /tmp/wtz_bench/gbo_ecoli.ovl16 -fo /tmp/wtz_bench/gbo_gpu.ovl -9 /tmp/wtz_bench/gbo_gpu.pairs 2> $O/wtgbo_gpu.err ) 2> $O/wtgbo_gpu.time
grep -E "candidates|new overlaps|wtgbo-mi355x" $O/wtgbo_gpu.err; grep real $O/wtgbo_gpu.time
( time oracle/_ref/wtgbo_ref -t 32 -i $FA -j /tmp/wtz_bench/gbo_ecoli.ovl16 -fo /tmp/wtz_bench/gbo_ref32.ovl 2> $O/wtgbo_ref32.err ) 2> $O/wtgbo_ref32.time; echo "reference -t 32:"; grep real $O/wtgbo_ref32.time
if [ "$REF1" = "1" ]; then
( time oracle/_ref/wtgbo_ref -t 1 -i $FA -j /tmp/wtz_bench/gbo_ecoli.ovl16 -fo /tmp/wtz_bench/gbo_ref1.ovl -9 /tmp/wtz_bench/gbo_ref1.pairs 2> $O/wtgbo_ref1.err ) 2> $O/wtgbo_ref1.time; echo "reference -t 1:"; grep real $O/wtgbo_ref1.time
grep -E "candidates|new overlaps" $O/wtgbo_ref1.err | head -12
md5sum /tmp/wtz_bench/gbo_gpu.ovl /tmp/wtz_bench/gbo_ref1.ovl /tmp/wtz_bench/gbo_gpu.pairs /tmp/wtz_bench/gbo_ref1.pairs | tee $O/md5.txt
fi
wc -l /tmp/wtz_bench/gbo_gpu.ovl
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gbo -o gbo -- $R/bin/wtgbo -i $FA -j /tmp/wtz_bench/gbo_ecoli.ovl16 -fo /tmp/wtz_bench/gbo_gpu2.ovl > $O/trace_gbo.log 2>&1
cd $R
python tools/summarize_profiles.py $O $O/summary
find $O -name "*kernel_trace.csv" -size +8M -delete
ls $O/summary
