#!/bin/bash
# parity subset + timing lines (both workloads, both engines); WTZ_CAND_STREAM=0 lines for comparison
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fresh" 2>&1 | tail -4
bash tools/gpu_r02_exp4.sh r02k
echo "== sorting form of the candidate kernel (WTZ_CAND_STREAM=0)"
for f in ecoli yeast100; do WTZ_CAND_STREAM=0 bin/wtzmo -i /tmp/$f.fa -fo /tmp/y.ovl -k 16 -s 200 -m 0.6 2>&1 | grep -E "kernel ms" | cut -c1-120; done
