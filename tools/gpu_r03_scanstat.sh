#!/bin/bash
# K_pair window scans (potential_paired_kmers_windows, hzm_aln.h:410-578): how many find NO window on the candidate axis, and what the exact
# early exit (adjacent-bin bound, wtz_window.h) saves.  Phase-profiler builds: slot 21 = scans past the cheap gates, 22 = their matches, 23 = scans whose
# sweep finds nothing (-DWTZ_EXP_CNT_EMPTY).  E. coli-shape input, zmo; then the shipped build with and without the exit on the same input + wtgbo.
TAG=${1:-r03scan}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
mkdir -p /tmp/wtz_bench
FA=/tmp/wtz_bench/gbo_ecoli.fa
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
print(bench.gen_reads("$FA", 4600000, 25.0, 11))
PY
cp smartdenovo_amd/libwtzmo_hip.so /tmp/lib_keep.so
for V in "-DWTZ_NO_SCAN_PRECHECK" ""; do
  WTZ_HIPCC_FLAGS="-DWTZ_PROFILE -DWTZ_EXP_CNT_EMPTY $V" python -c "import __graft_entry__ as g; g.build_product(force=True)" > $O/build.log 2>&1 || tail -5 $O/build.log
  echo "== profile build [$V]"
  ( time env WTZ_PROFILE_PAIR=1 bin/wtzmo -i $FA -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 ) > $O/zmo.err 2>&1
  grep -E "phase-profile" $O/zmo.err | tail -1 | tr ' ' '\n' | grep -E "^(16|17|18|19|20|21|22|23):" | tr '\n' ' '; echo; md5sum /tmp/e.ovl
done
for V in "-DWTZ_NO_SCAN_PRECHECK" ""; do
  WTZ_HIPCC_FLAGS="$V" python -c "import __graft_entry__ as g; g.build_product(force=True)" > $O/build.log 2>&1 || tail -5 $O/build.log
  echo "== shipped build [$V]"
  bin/wtzmo -i $FA -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 --repeat 3 2>&1 | grep -E "kernel ms" | tail -1 | cut -c1-200; md5sum /tmp/e.ovl
  bin/wtzmo -i $FA -fo /tmp/d.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 --repeat 3 2>&1 | grep -E "kernel ms" | tail -1 | cut -c1-200; md5sum /tmp/d.ovl
  cut -f1-16 /tmp/e.ovl > /tmp/e.ovl16
  ( time bin/wtgbo -i $FA -j /tmp/e.ovl16 -fo /tmp/g.ovl ) 2>&1 | grep -E "kernel ms|real"; md5sum /tmp/g.ovl
done
cp /tmp/lib_keep.so smartdenovo_amd/libwtzmo_hip.so
