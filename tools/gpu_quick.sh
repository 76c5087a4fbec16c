#!/bin/bash
# quick E. coli-shape run of both engines: timing lines, stats rows, md5 of the outputs
mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY' > gpurun_out/gen.log 2>&1
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
echo "== zmo"; WTZ_PROFILE_PAIR=${WTZ_PROFILE_PAIR:-0} bin/wtzmo --pool-gb 60 --stats /tmp/z.stats $WTZ_EXTRA -i /tmp/ecoli.fa -fo /tmp/e.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "records,|batches|kernel ms|host seconds|wall seconds|split|pair-profile|phase-profile|ext-profile|gap-profile|align-profile"; md5sum /tmp/e.ovl; cat /tmp/z.stats
echo "== dmo"; WTZ_PROFILE_PAIR=${WTZ_PROFILE_PAIR:-0} bin/wtzmo --pool-gb 40 --stats /tmp/d.stats $WTZ_EXTRA -i /tmp/ecoli.fa -fo /tmp/d.ovl -k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000 2>&1 | grep -E "records,|batches|kernel ms|host seconds|wall seconds|split|pair-profile|phase-profile|ext-profile|gap-profile|align-profile"; md5sum /tmp/d.ovl; cat /tmp/d.stats
echo "expect zmo 3c46e34fd78ef9667fd72ad151100b59 dmo aaeb67d219a9225cbfb7898f5c970983"
