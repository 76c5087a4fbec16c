#!/bin/bash
# K-sw3 multi-wave policy sweep on the E. coli-shape set: "min:top" pairs (WTZ_SW_MW_MIN : WTZ_SW_MW_TOP)
mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
for mt in ${WTZ_MW_SWEEP:-0:0 256:100000}; do
	echo "== min:top $mt"
	WTZ_SW_MW_MIN=${mt%%:*} WTZ_SW_MW_TOP=${mt##*:} bin/wtzmo --pool-gb 60 -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "kernel ms|records|ext-profile|mw-profile" | cut -c1-170
	md5sum /tmp/e.ovl
done
