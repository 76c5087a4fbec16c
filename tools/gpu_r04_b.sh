#!/bin/bash
# round 4, second GPU run: the lane-interleaved trace + writer threads (parity first), then K-sw3 placement / occupancy experiments and the split z-index cost
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-r04b}; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_dp_forms.py tests/test_gpu_parity.py tests/test_binary_handoff.py -m gpu -x -q ) > $O/pytest_fast.log 2>&1; tail -3 $O/pytest_fast.log
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1"
run(){ tag=$1; shift; ( "$@" ) > $O/$tag.json 2> $O/$tag.err; python3 - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d['kernel_ms_last_step']
    print("%-22s %.3f s/step %.2f Gbp/s parity %s | pairs %.0f sw1 %.0f stitch %.0f ksw3 %.0f ksw2 %.0f" % (sys.argv[2], d['ms_per_step']/1e3, d['value'], d['parity'].get('match'), k['pairs'], k['winalign'], k['stitch'], k['ksw3_wave'], k['ksw2_gap']))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
run base $B
grep "host seconds\|commit sections" $O/base.err | tail -2
for mw in 2048 100000000; do WTZ_SW_MW_MIN=$mw run base_mw$mw $B; done
for mw in 512 2048 100000000; do WTZ_SW_MW_MIN=$mw run occ2_mw$mw tools/with_variant.sh occ2 $B; done
WTZ_SW_MW_MIN=0 run base_mw0_allone $B
run dmo python bench.py --no-cpu-baseline --steps 2 --warmup 1 --engine dmo
# the z-mer index: all reads / the candidate side of one residue class of 8 / the queries of a 4096-query batch (second call of each: arrays recycled)
python3 - > $O/zindex_split.txt 2>&1 <<'PY'
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from smartdenovo_amd import hipabi
lib = C.CDLL(hipabi.LIB_PATH)
P = hipabi.Params.defaults()
fa = "/tmp/wtz_bench/reads_G12000000_c100_s29.fa"
names, seqs = [], []
for ln in open(fa):
    if ln[0] == '>': names.append(ln[1:].strip())
    else: seqs.append(ln.strip())
order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
seqs = [seqs[i] for i in order]
lens = np.array([len(s) for s in seqs], dtype=np.uint32); off = np.zeros(len(seqs), dtype=np.uint64); off[1:] = np.cumsum(lens[:-1])
txt = "".join(seqs).encode()
ctx = C.c_void_p()
lib.wtz_ctx_create.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
assert lib.wtz_ctx_create(0, C.byref(P), 8 << 30, C.byref(ctx)) == 0
lib.wtz_upload_reads_ascii.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
assert lib.wtz_upload_reads_ascii(ctx, txt, len(txt), off.ctypes.data, lens.ctypes.data, len(seqs), 0, None) == 0
lib.wtz_zindex_build.argtypes = [C.c_void_p]; lib.wtz_zindex_build_subset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; lib.wtz_zindex_build_queries.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
def t(f, *a):
    r = []
    for _ in range(3):
        t0 = time.perf_counter(); assert f(*a) == 0; r.append(time.perf_counter() - t0)
    return ["%.1f ms" % (x * 1e3) for x in r]
n = len(seqs)
print("reads", n, "bases", len(txt))
print("all reads          ", t(lib.wtz_zindex_build, ctx))
for N in (2, 4, 8):
    ids = np.arange(0, n, N, dtype=np.uint32)
    print("residue class 0 of %d" % N, t(lib.wtz_zindex_build_subset, ctx, ids.ctypes.data, len(ids)))
    for nq in (256, 1024, 4096):
        q = np.arange(20000, 20000 + nq, dtype=np.uint32)
        print("   + queries of a batch of %d:" % nq, t(lib.wtz_zindex_build_queries, ctx, q.ctypes.data, nq))
PY
cat $O/zindex_split.txt
