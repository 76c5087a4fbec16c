#!/bin/bash
# f2 at scale on the GPU box: bin/wtzmo on the configs[2] input (481 648 records with their CIGARs) -> reference wtobt (oracle/_ref) -> reference wtext with all host
# threads and bin/wtext on the same files; the two outputs must be the same bytes.  Also the GPU tests of tests/test_wtext.py.
# usage: tools/gpu_wtext_scale.sh <tag> [ecoli]     outputs under gpurun_out/<tag>/
TAG=${1:-wtext}; WL=${2:-yeast100}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_wtext.py -m gpu -x -q ) > $O/pytest_wtext.log 2>&1; tail -4 $O/pytest_wtext.log | head -2
T=/tmp/wtz_bench; mkdir -p $T
if [ "$WL" = ecoli ]; then G=4600000; C=25; S=11; else G=12000000; C=100; S=29; fi
FA=$T/reads_G${G}_c${C}_s${S}.fa
python3 -c "import bench; print(bench.gen_reads('$FA', $G, $C, $S))"
NT=$(nproc)
( time timeout 900 bin/wtzmo -i $FA -fo $T/x.ovl -k 16 -s 200 -m 0.6 ) > $O/wtzmo.out 2> $O/wtzmo.err; grep "records," $O/wtzmo.err | tail -1; ls -la $T/x.ovl | awk '{print $5, "bytes of overlaps"}'
( time timeout 900 oracle/_ref/wtobt_ref -i $FA -j $T/x.ovl -fo $T/x.obt -m 0.6 -c 2 ) > $O/wtobt.out 2> $O/wtobt.err; grep real $O/wtobt.err; wc -l < $T/x.obt
( time timeout 1800 oracle/_ref/wtext_ref -t $NT -i $FA -j $T/x.ovl -b $T/x.obt -fo $T/x_ref.ext ) > $O/wtext_ref.out 2> $O/wtext_ref.err; echo "reference wtext -t $NT:"; grep real $O/wtext_ref.err
( time timeout 1800 bin/wtext -t $NT -i $FA -j $T/x.ovl -b $T/x.obt -fo $T/x_gpu.ext ) > $O/wtext_gpu.out 2> $O/wtext_gpu.err; echo "bin/wtext:"; grep "real\|host seconds\|extension kernels\|records written" $O/wtext_gpu.err
# the reference writes its batches in the order its workers finish: compare as sets of lines when the bytes differ, and say which it was
A=$(md5sum < $T/x_ref.ext); B=$(md5sum < $T/x_gpu.ext)
if [ "$A" = "$B" ]; then echo "PARITY bytes equal ($A)"; else
  SA=$(sort $T/x_ref.ext | md5sum); SB=$(sort $T/x_gpu.ext | md5sum)
  if [ "$SA" = "$SB" ]; then echo "PARITY same records, the reference's thread order differs (sorted md5 $SA)"; else echo "PARITY MISMATCH"; fi
fi | tee $O/parity.txt
wc -l $T/x_ref.ext $T/x_gpu.ext | head -2
( time timeout 1800 oracle/_ref/wtext_ref -t 1 -i $FA -j $T/x.ovl -b $T/x.obt -fo $T/x_ref1.ext ) > /dev/null 2> $O/wtext_ref_t1.err; echo "reference wtext -t 1:"; grep real $O/wtext_ref_t1.err; cmp $T/x_ref1.ext $T/x_gpu.ext && echo "PARITY -t 1 bytes equal" | tee -a $O/parity.txt
rm -f $T/x.ovl $T/x_ref.ext $T/x_gpu.ext $T/x_ref1.ext
