#!/bin/bash
export TMPDIR=/tmp
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11)
PY
WTZ_HIPCC_FLAGS="-DWTZ_PROFILE" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
WTZ_PROFILE_PAIR=1 bin/wtzmo -n --pool-gb 80 -i /tmp/ecoli.fa -fo /tmp/en.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "kernel ms|phase-profile" | grep -v "0:0.0 1:0.0" | sed 's/.* 23:[0-9.]* //'
