#!/bin/bash
# Where the seed-lookup workgroups spend their time: the -DWTZ_PROFILE_CAND build (tools/build_variant.sh cprof -DWTZ_PROFILE_CAND), one step of each engine.
# slots (wtz_task_candidates_wg): 0 k-mer walk, 1 probe + histogram, 2 bin scan + allocation, 3 scatter, 4 bucket bounds, 5 copy-in + sort, 6 intervals + group fold, 7 heap tail;
# counts: 8 queries, 9 sampled k-mers, 10 tuples, 11 buckets, 12 groups >= -d, 13 heap entries
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-cprof}; mkdir -p $O
export TMPDIR=/tmp
cd $R
for e in zmo dmo; do
  WTZ_PROFILE_PAIR=1 timeout 600 tools/with_variant.sh ${2:-cprof} python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-verify --engine $e > $O/cand_$e.json 2> $O/cand_$e.err
  grep "cand-profile" $O/cand_$e.err | python3 -c "
import sys,re
S=[0.0]*16
for l in sys.stdin:
    for m in re.finditer(r' (\d+):([0-9.]+)', l): S[int(m.group(1))]+=float(m.group(2))
print('$e', ' '.join('%d:%.1f' % (k, v) for k, v in enumerate(S)))"
  grep "kernel ms" $O/cand_$e.err | tail -1
done
