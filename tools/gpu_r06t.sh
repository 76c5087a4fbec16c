#!/bin/bash
# round 6: resident wavefronts of the isolated K-sw3 kernels (no scratch) on 40 000 jobs: SQ_WAVE_CYCLES x 4 / (SQ_BUSY_CYCLES / 32)
TAG=${1:-r06t}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for f in 5 7; do
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/f$f -o x -- python $R/tools/ubench/ksw3_bench.py --forms $f --reps 1 --jobs 40000 > $O/f$f.log 2>&1
done
cd $R
python3 - $O <<'PY'
import csv,sys,glob,collections
O=sys.argv[1]
for f in (5,7):
    tot=collections.Counter()
    for fn in glob.glob(O+'/f%d/**/*counter_collection.csv'%f, recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'extjobs' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
    print('form',f,{k:int(v) for k,v in sorted(tot.items())}, 'resident', tot['SQ_WAVE_CYCLES']*4/(tot['SQ_BUSY_CYCLES']/32))
PY
find $O -name "*.csv" -size +1M -delete
