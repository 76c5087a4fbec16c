#!/bin/bash
# build the device library with different -D flags on the GPU box and time the E. coli-shape zmo run with each
# usage: tools/gpu_variants.sh "<flags A>" "<flags B>" ...     ("" = default build)
export TMPDIR=/tmp
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11)
PY
for f in "$@"; do
  WTZ_HIPCC_FLAGS="$f" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  for e in zmo ${WTZ_ENGINES}; do
    if [ $e = zmo ]; then A="-k 16 -s 200 -m 0.6"; else A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; fi
    bin/wtzmo --pool-gb 60 -i /tmp/ecoli.fa -fo /tmp/v.ovl $A 2>&1 | grep -E "records,|kernel ms|phase-profile"; md5sum /tmp/v.ovl | cut -c1-32
  done
done
