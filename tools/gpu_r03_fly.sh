#!/bin/bash
# BASELINE configs[3] shape on ONE device: 951 827 synthetic reads / 9.8 Gbp (140 Mbp iid genome x70), the query stripe -P 128 -p 0 against the
# FULL index, per-batch z-mer index; the .ovl must have the md5 of the reference's `wtzmo -t 1 -P 128 -p 0` (tests/golden/big_manifest.json).
TAG=${1:-r03fly}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
df -h /tmp | tail -1; free -g | head -2
( time WTZ_TEST_FLY=1 WTZ_TEST_KEEP_STDERR=$O timeout 3300 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k fly ) > $O/pytest_fly.log 2>&1
tail -5 $O/pytest_fly.log
grep -E "read bases|reads \(|index|records|kernel ms|host seconds|batches in|z-mer" $O/scale_fly70_zmo_P128p0.stderr.txt | tail -12 | cut -c1-300
