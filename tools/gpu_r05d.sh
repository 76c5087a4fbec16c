#!/bin/bash
# round 5: is the bump pool's single counter what the short-lived waves wait for?  64 counters (experimental build) against one: K-sw3 in isolation, the whole step
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base shards shards_rows1; do
  echo "== $v" >> $O/variants.txt
  timeout 300 tools/with_variant.sh $v python tools/ubench/ksw3_bench.py --forms 1,5,2,0 --no-compare --reps 2 >> $O/variants.txt 2>> $O/variants.err
done
cat $O/variants.txt
for v in base shards; do
  timeout 600 tools/with_variant.sh $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  grep "kernel ms" $O/bench_$v.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_$v.json').read().strip().split('\n')[-1])
print('$v', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d.get('parity'))
"
done
for v in base shards; do
  timeout 600 tools/with_variant.sh $v python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dmo_$v.json 2> $O/bench_dmo_$v.err
  grep "kernel ms" $O/bench_dmo_$v.err | tail -1
  python3 -c "
import json
d=json.loads(open('$O/bench_dmo_$v.json').read().strip().split('\n')[-1])
print('dmo $v', d['ms_per_step'], d['value'], d.get('parity'))
"
done
