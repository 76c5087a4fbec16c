#!/bin/bash
# round 6: SQ counters of the fused K-sw3 launches inside a configs[2] step, packed (default) and 32-bit (WTZ_EXT_PK=0)
TAG=${1:-r06s}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
p=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_WAIT_ANY SQ_IFETCH"; do
  p=$((p+1))
  for f in 1 0; do
    WTZ_EXT_PK=$f timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/p${p}_pk$f -o x -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/p${p}_pk$f.log 2>&1
  done
done
cd $R
python3 - $O <<'PY'
import csv,sys,glob,collections
O=sys.argv[1]
for f in (1,0):
    tot=collections.defaultdict(collections.Counter)
    for fn in glob.glob(O+'/p*_pk%d/**/*counter_collection.csv'%f, recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'stitch_ext' in r['Kernel_Name']: tot[r['Kernel_Name'][:32]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in tot.items(): print('pk',f,k,{a:int(b) for a,b in sorted(v.items())})
PY
find $O -name "*.csv" -size +1M -delete
