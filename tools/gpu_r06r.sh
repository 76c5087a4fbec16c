#!/bin/bash
# round 6: SQ counters of the packed K-sw3 (form 7) against the 32-bit frame form (form 5) on 2 000 jobs (one wavefront per SIMD: the latency regime of the step's launches)
TAG=${1:-r06r}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
p=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU"; do
  p=$((p+1))
  for f in 5 7; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/p${p}_f$f -o x -- python $R/tools/ubench/ksw3_bench.py --forms $f --reps 1 --jobs 2000 > $O/p${p}_f$f.log 2>&1
  done
done
cd $R
python3 - $O <<'PY'
import csv,sys,glob,collections
O=sys.argv[1]
for f in (5,7):
    tot=collections.Counter()
    for fn in glob.glob(O+'/p*_f%d/**/*counter_collection.csv'%f, recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'extjobs' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
    print('form',f,{k:int(v) for k,v in sorted(tot.items())})
PY
find $O -name "*.csv" -size +2M -delete
