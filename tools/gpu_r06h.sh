#!/bin/bash
# round 6: seed-lookup phase clock at the configs[3] shape with the 8-bit sketch; FETCH_SIZE / WRITE_SIZE by access width; idle gaps of a configs[2] step; commit with flush prefetch
TAG=${1:-r06h}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
line(){ python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d.get('kernel_ms_last_step',{})
    print("%-10s n %d %.3f s/step %.2f Gbp/s parity %s frac %.4f seed %.4f | %s | host %s" % (sys.argv[2], d['n_gpus'], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), d['roofline'].get('frac') or 0, (d.get('roofline_seed') or {}).get('frac') or 0, {a:round(b) for a,b in k.items()}, d.get('host_seconds_last_step')))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
}
( time timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_zmo.json 2> $O/bench_zmo.err; line $O/bench_zmo.json zmo; grep "helper threads" $O/bench_zmo.err | tail -1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/fw_$c -o fw -- $R/tools/ubench/fetch_width > $O/fw_$c.log 2>&1; python3 $R/tools/analysis/pmc_by_kernel.py $O/fw_$c > $O/fetch_width_$c.txt 2>&1; cat $O/fetch_width_$c.txt; done
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_zmo -o t -- python $R/bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/trace_zmo.log 2>&1
python3 $R/tools/analysis/idle_gaps.py $O/trace_zmo > $O/idle_gaps_zmo.txt 2>&1; head -4 $O/idle_gaps_zmo.txt | cut -c1-600
cd $R
( time WTZ_PROFILE_PAIR=1 timeout 2400 tools/with_variant.sh cprof python bench.py --workload fly70 --steps 1 --warmup 0 --no-cpu-baseline --no-verify ) > $O/cand_fly70.json 2> $O/cand_fly70.err
grep "cand-profile" $O/cand_fly70.err | python3 -c "
import sys,re
S=[0.0]*16
for l in sys.stdin:
    for m in re.finditer(r' (\d+):([0-9.]+)', l): S[int(m.group(1))]+=float(m.group(2))
print('fly70 cand slots', ' '.join('%d:%.1f' % (k, v) for k, v in enumerate(S)))"
grep "kernel ms" $O/cand_fly70.err | tail -1
grep "cand-profile" $O/cand_fly70.err | head -3 | cut -c1-300
rm -f /tmp/wtz_bench/reads_G140000000_*
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*copy_trace.csv" -size +1M -delete; find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*.db" -delete
