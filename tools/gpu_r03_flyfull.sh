#!/bin/bash
# BASELINE configs[3] shape as ONE whole job on ONE device: 951 827 synthetic reads / 9.8 Gbp (140 Mbp iid genome x70), per-batch z-mer index.
# First the stripe test (-P 128 -p 0 == reference md5; it generates the 10 GB FASTA and keeps it), then every query against the full index; the
# ~80 GB of records go to /dev/null (no reference to compare the whole file with: `wtzmo -t 1` would take days; the stripe pins parity).
TAG=${1:-r03flyfull}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
df -h /tmp | tail -1; free -g | head -2
( time WTZ_TEST_KEEP_FLY=1 WTZ_TEST_KEEP_STDERR=$O timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k fly ) > $O/pytest_fly.log 2>&1; tail -3 $O/pytest_fly.log
FLY=$(ls /tmp/wtz_bench/reads_G140000000_c70_s53.fa 2>/dev/null)
if [ -n "$FLY" ]; then
  ( time timeout 2400 bin/wtzmo -i $FLY -fo /dev/null -C -k 16 -s 200 -m 0.6 --stats $O/fly_full.stats ) > $O/fly_full.log 2>&1
  grep -E "reads \(|packed|records|host seconds|batches in|kernel ms|real|z-mer" $O/fly_full.log | cut -c1-300
  cat $O/fly_full.stats | cut -c1-200
  rm -f /tmp/wtz_bench/reads_G140000000_c70_s53.fa*
else echo "no fly input kept"; ls /tmp/wtz_bench | head; fi
