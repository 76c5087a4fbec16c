#!/bin/bash
# round 5, first measurement: instruction rates of candidate K-sw3 cell ops, the geometry of the K-sw3 jobs of a configs[2] step (dump for tools/ubench/ksw3_bench.py),
# the counters rocprofv3 offers for instruction fetch, band-class launches (one CMAX instantiation per kernel object) vs the merged kernel
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 tools/ubench/valu_int32 > $O/valu.txt 2>&1
(rocprofv3 -L 2>&1 || rocprofv3 --list-avail 2>&1) | grep -i -E "^\s*(name|counter)?.*(IFETCH|ICACHE|SQC_|SQ_INST_LEVEL|SQ_IFETCH|INSTS_|SQ_WAIT|SQ_BUSY|SQ_ACTIVE)" | head -150 > $O/counters.txt
WTZ_PROFILE_PAIR=1 WTZ_EXT_DUMP=$O/extjobs.bin timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_profile.json 2> $O/bench_profile.err
grep "ext-profile" $O/bench_profile.err | head -200 > $O/ext_profile.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
WTZ_EXT_SPLIT=2 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_split2.json 2> $O/bench_split2.err
WTZ_EXT_SPLIT=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_split1.json 2> $O/bench_split1.err
WTZ_SW_MW_MIN=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_nomw.json 2> $O/bench_nomw.err
ls -la $O
