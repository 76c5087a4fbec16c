#!/bin/bash
# K-sw3 on-device cross-check (wave/register kernels vs the scalar body) on the 600-read smoke set + tiny goldens, then the quick E. coli run
export TMPDIR=/tmp
WTZ_SW_CHECK=1 bin/wtzmo -i tests/golden/tiny.fa.gz -fo /tmp/t.ovl -k 16 -s 200 -m 0.6 2>&1 | tail -3
python - <<'PY'
import gzip,hashlib,json
m=json.load(open('tests/golden/manifest.json'))
c=[x for x in (m['cases'] if 'cases' in m else m) if (x.get('name') if isinstance(x,dict) else x)=='zmo']
print('golden entry:', str(c)[:200])
PY
md5sum /tmp/t.ovl
bash tools/gpu_quick.sh 2>&1 | grep -B12 "== dmo" | grep -E "kernel ms|records|e.ovl"
