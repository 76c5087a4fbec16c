#!/bin/bash
# quick experiments on configs[2] (zmo): four-wave K-sw3 share, two worker contexts, K_pair phase ticks (E. coli shape)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench
print(bench.gen_reads('/tmp/yeast100.fa',12000000,100.0,29)); print(bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11))
PY
A="-k 16 -s 200 -m 0.6"
run(){ tag=$1; shift; echo "== $tag"; ( time "$@" bin/wtzmo -i /tmp/yeast100.fa -fo /tmp/y.ovl $A $EXTRA ) > $O/$tag.err 2>&1; grep -E "records,|kernel ms|batches|real|wall seconds|host seconds" $O/$tag.err | cut -c1-260; md5sum /tmp/y.ovl | cut -c1-12; }
EXTRA=""
run base env
run mwtop0 env WTZ_SW_MW_TOP=0
run mwtop256 env WTZ_SW_MW_TOP=256
run mwtop1024 env WTZ_SW_MW_TOP=1024
run mwmin2048 env WTZ_SW_MW_MIN=2048
EXTRA="--workers 2"
run workers2 env
EXTRA="--batch 4096"
run batch4096 env
echo "== ecoli pair profile"
WTZ_PROFILE_PAIR=1 bin/wtzmo -i /tmp/ecoli.fa -fo /tmp/e.ovl $A 2>&1 | grep -E "pair-profile\] n=|align-profile|winalign-profile|gap-profile|ext-profile\] [0-9]" | cut -c1-330 | tail -24
