#!/bin/bash
# Diagnostic (results are WRONG by construction): what bounds K-sw3 -- the trace stores or the traceback?  Same run with libraries built
# -DWTZ_EXP_NOTRACE (trace computed, not stored), -DWTZ_EXP_NOTB (no traceback), both.
TAG=${1:-r03diag}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
FA=$(ls /tmp/wtz_bench/reads_G12000000_c100_s29.fa)
cp smartdenovo_amd/libwtzmo_hip.so /tmp/lib_orig.so
for v in orig NOTRACE NOTB BOTH; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so smartdenovo_amd/libwtzmo_hip.so; else cp tools/exp_libs/libwtzmo_hip_$v.so smartdenovo_amd/libwtzmo_hip.so; fi
  timeout 300 bin/wtzmo -i $FA -fo /tmp/wtz_bench/diag.ovl -k 16 -s 200 -m 0.6 --repeat 2 2> $O/diag_$v.err
  echo "== $v rc $?"; grep -E "records|kernel ms" $O/diag_$v.err | tail -2 | cut -c1-220
done
cp /tmp/lib_orig.so smartdenovo_amd/libwtzmo_hip.so; rm -f /tmp/wtz_bench/diag.ovl*
