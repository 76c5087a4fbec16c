#!/bin/bash
# quick check after the finer-bin scan exit: E. coli-shape zmo / wtgbo kernel times + md5s, wtgbo start-up against the scratch pool size
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; mkdir -p /tmp/wtz_bench
FA=/tmp/wtz_bench/gbo_ecoli.fa
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
print(bench.gen_reads("$FA", 4600000, 25.0, 11))
PY
bin/wtzmo -i $FA -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 --repeat 3 2>&1 | grep -E "kernel ms|packed" | tail -2 | cut -c1-220; md5sum /tmp/e.ovl
cut -f1-16 /tmp/e.ovl > /tmp/e.ovl16
for pg in 0 16 4; do
  ( time bin/wtgbo -i $FA -j /tmp/e.ovl16 -fo /tmp/g.ovl --pool-gb $pg ) 2>&1 | grep -E "kernel ms|real" | cut -c1-200; md5sum /tmp/g.ovl
done
