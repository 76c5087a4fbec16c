#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04q}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_a-z0-9]*\|TA_[A-Z_a-z0-9]*\|TCC_[A-Z_a-z0-9]*\|GRBM_[A-Z_a-z0-9]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o dmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --engine dmo > $O/pmc_$tag.log 2>&1 || tail -3 $O/pmc_$tag.log
done
cd $R
python3 - $O <<'PY'
import csv,glob,sys,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/pmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        m=re.search(r'<(K_\w+)',k); k=m.group(1) if m else k[:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k in ['K_pair_dm','K_candidates_wg']:
    print(k,{c:'%.3g'%v for c,v in sorted(agg[k].items())})
PY
find $O -name "*counter_collection.csv" -size +8M -delete
