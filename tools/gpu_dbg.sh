#!/bin/bash
mkdir -p gpurun_out
G=tests/golden
for v in 0 1; do
 echo "== WTZ_SYNC_ALLOC=$v"
 WTZ_SYNC_ALLOC=$v timeout 300 bin/wtzmo --workers 1 -i $G/tiny.fa.gz -fo /tmp/t$v.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "index 1|records,"
 md5sum /tmp/t$v.ovl
done > gpurun_out/dbg.log 2>&1
echo "== workers 2 sync alloc" >> gpurun_out/dbg.log
WTZ_SYNC_ALLOC=1 timeout 300 bin/wtzmo --workers 2 --first-batch 8 --batch 32 -i $G/tiny.fa.gz -fo /tmp/t2.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "index 1|records,|batches" >> gpurun_out/dbg.log; md5sum /tmp/t2.ovl >> gpurun_out/dbg.log
cat gpurun_out/dbg.log
