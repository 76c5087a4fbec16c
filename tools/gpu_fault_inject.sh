#!/bin/bash
# fault injection into the device scratch pools (WTZ_POOL_FAIL_AT / WTZ_TPOOL_FAIL_AT = the n-th request of every stage call fails):
# every run must either finish with the golden output or exit(1) loudly - a GPU memory fault names the stage it happened in
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-fi}; mkdir -p $O
cd $R
IN=${WTZ_FI_INPUT:-tests/golden/tiny.fa.gz}
if [ "$IN" = repeat ]; then
python - <<'PY'
import sys; sys.path.insert(0,'.')
from smartdenovo_amd import synth
n,s = synth.synth_reads(150000, 10, seed=123, mean_len=9000.0, min_len=1000, repeats=True)
synth.write_fasta('/tmp/fi_rep.fa', n, s)
PY
IN=/tmp/fi_rep.fa
fi
declare -A ENG=( [zmo]="-k 16 -s 200 -m 0.6" [dmo]="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000" [zmo_n]="-k 16 -s 200 -m 0.6 -n" )
bad=0
for e in zmo dmo zmo_n; do
  for var in WTZ_POOL_FAIL_AT WTZ_TPOOL_FAIL_AT; do
    [ $e = dmo ] && [ $var = WTZ_TPOOL_FAIL_AT ] && continue
    for n in 1 2 3 4 6 9 14 20 30 45 70 100 150 230 350 500 750 1100 1700 2500 4000 6000 9000 14000 20000 30000 45000; do
      env $var=$n WTZ_STAGE_TRACE=1 timeout 120 bin/wtzmo --pool-mb 4096 -i $IN -fo /tmp/fi.ovl ${ENG[$e]} > /tmp/fi.err 2>&1; rc=$?
      if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then bad=$((bad+1)); echo "FAULT $e $var=$n rc=$rc last stage: $(grep '^\[stage\]' /tmp/fi.err | tail -1); $(grep -m1 -E 'Memory access|fault' /tmp/fi.err | cut -c1-120)"; cp /tmp/fi.err $O/fault_${e}_${var}_$n.err; fi
    done
  done
done
echo "fault-injection sweep: $bad faulting runs"
