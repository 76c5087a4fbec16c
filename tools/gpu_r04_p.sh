#!/bin/bash
# PMC pass over one dmo step: what bounds K_pair_dm (issue, LDS, memory)?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04p}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o dmo -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --engine dmo > $O/pmc_$tag.log 2>&1
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
for k in sorted(agg,key=lambda k:-agg[k].get('SQ_WAVE_CYCLES',0))[:4]:
    print(k,{c:'%.3g'%v for c,v in sorted(agg[k].items())})
PY
find $O -name "*counter_collection.csv" -size +8M -delete
