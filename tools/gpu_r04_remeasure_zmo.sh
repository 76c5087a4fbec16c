#!/bin/bash
# The zmo half of tools/gpu_r04_final.sh (driver's bench command, rocprofv3 kernel statistics, PMC passes) - run after the last host-side change of round 4 (buffer reuse in the
# index builds) so that the headline line and its traffic figures carry the build id of the committed sources.  usage: tools/gpu_r04_remeasure_zmo.sh <tag>
TAG=${1:-r04final3}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 -c "
import json; d=json.loads(open('$O/bench_driver_cmd.json').read().strip().split('\n')[-1]); print('driver cmd: %.3f s/step %.2f Gbp/s parity %s frac %.4f'%(d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline']['frac']), d['kernel_ms_last_step'])"
grep "records," $O/bench_driver_cmd.err | cut -c60-140 | sort | uniq -c | sort -rn | head -n 4
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_zmo -o zmo -- python $R/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 > $O/trace_zmo.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_SQ -o zmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_SQ.log 2>&1
cd $R
mkdir -p $O/split_zmo; for d in trace_zmo pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do [ -d $O/$d ] && ln -s $O/$d $O/split_zmo/$d; done
python tools/summarize_profiles.py $O/split_zmo $O/summary_zmo
mkdir -p $O/summary
KID=$(python3 -c "import bench; print(bench.kernel_source_id())")
[ -f $O/summary_zmo/pmc_per_kernel.csv ] && { cp $O/summary_zmo/pmc_per_kernel.csv $O/summary/r04_yeast100_zmo_pmc_per_kernel.csv; echo "{\"kernel_source_id\": \"$KID\", \"command\": \"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> -- python bench.py --engine zmo --steps 1 --warmup 0\"}" > $O/summary/r04_yeast100_zmo_pmc_per_kernel.csv.meta.json; }
[ -f $O/summary_zmo/trace_zmo_kernel_stats.csv ] && cp $O/summary_zmo/trace_zmo_kernel_stats.csv $O/summary/r04_yeast100_zmo_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
rm -rf $O/split_zmo; ls $O/summary
