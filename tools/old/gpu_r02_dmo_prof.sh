#!/bin/bash
# configs[2] dmo with the phase profiler build: where does K_pair spend its 10 s?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29))
PY
WTZ_HIPCC_FLAGS="-DWTZ_PROFILE" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
( time env WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/yeast100.fa -fo /tmp/d.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 ) > /tmp/d.err 2>&1
grep -E "phase-profile" /tmp/d.err | cut -c1-700
grep -E "pair-profile\] n=" /tmp/d.err | awk '{split($8,a,"/"); m+=a[1]; s+=a[2]; w+=a[3]; t+=a[4]; n+=substr($2,3)} END{print "pairs", n, "kticks match/sort/denoise/total", m, s, w, t}'
grep -E "tier" /tmp/d.err | awk '{n[$4]+=$8; t[$4]+=$(NF-1)} END{for(k in n) print "tier", k, "pairs", n[k], "ms", t[k]}'
grep -E "K_pair first" /tmp/d.err | awk '{n+=$5; t+=$(NF-1)} END{print "first launch pairs", n, "ms", t}'
grep -E "records,|kernel ms" /tmp/d.err | cut -c1-200
