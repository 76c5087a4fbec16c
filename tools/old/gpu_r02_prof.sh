#!/bin/bash
# Where does a configs[2] step go inside K_pair / K_winalign / K-sw3?  (1) per-pair tick sums of the shipped build, (2) the phase profiler build (-DWTZ_PROFILE).
TAG=${1:-r02p}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import bench
t=time.time(); print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29), 'gen %.1fs' % (time.time()-t))
PY
A="-k 16 -s 200 -m 0.6"
if [ -z "$PROF_ONLY" ]; then
echo "== shipped build, WTZ_PROFILE_PAIR=1"
( time timeout 600 env WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/yeast100.fa -fo /tmp/y.ovl $A ) 2> $O/ship.err
grep -E "pair-profile\] n=|winalign-profile|gap-profile|ext-profile|kernel ms|real" $O/ship.err | cut -c1-400 | tail -40
md5sum /tmp/y.ovl
fi
echo "== phase profiler build"
WTZ_HIPCC_FLAGS="-DWTZ_PROFILE" python -c "import __graft_entry__ as g; g.build_product(force=True)" > $O/build.log 2>&1 || tail -5 $O/build.log
( time timeout 900 env WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/yeast100.fa -fo /tmp/y2.ovl $A ) 2> $O/prof.err
grep -E "phase-profile|kernel ms|real" $O/prof.err | cut -c1-1500 | tail -30
md5sum /tmp/y2.ovl
