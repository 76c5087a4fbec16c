#!/bin/bash
# build the device library with different -D flags on the GPU box and time the configs[2] zmo run with each
# usage: tools/gpu_variants_big.sh "<flags A>" "<flags B>" ...     ("" = default build)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29)
PY
WTZ_ARGS=${WTZ_ARGS:--k 16 -s 200 -m 0.6}
for f in "$@"; do
  WTZ_HIPCC_FLAGS="$f" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  bin/wtzmo --repeat 2 -i /tmp/yeast100.fa -fo /tmp/v.ovl $WTZ_ARGS 2>&1 | grep -E "records,|kernel ms" | tail -2 | cut -c1-200; md5sum /tmp/v.ovl | cut -c1-32
done
