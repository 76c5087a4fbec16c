"""f1, function level: align_hzmaux (hzm_aln.h:1684-1775) — the pair routine of wtgbo — three ways on the same seeded read pairs:
the REAL reference routine (oracle/_ref/libref_shim.so), the oracle's restatement (oracle/ora_hzmaux.h), and the product's pair stages
in aux form (wtz_pairs_seed + wtz_pairs_align with params.aux_strand = 1) through the C-ABI — on the emulated device layer here,
on the MI355X under -m gpu.  Bit-exact: the ten integers of the kswx_t and every CIGAR word."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SHIM), reason="reference shim not built (needs /root/reference once)")

# zsize hz zwin zstep zovl zmax zvar w W ew rw M X I D E T   (wtgbo.c:385-411, 470-486)
WTGBO = [10, 1, 800, 0, 200, 100, 2, 50, 3200, 800, 50, 2, -5, -3, -3, -1, -50]
VARIANTS = {
    "wtgbo": (WTGBO, 0.6, 0),
    "refine": (WTGBO, 0.6, 1),
    "nohz_z12": ([12, 0, 600, 0, 150, 50, 1, 30, 1600, 400, 30, 3, -4, -2, -2, -2, -20], 0.65, 0),
    "refine_w20": ([10, 1, 800, 0, 200, 100, 2, 20, 3200, 800, 20, 2, -5, -3, -3, -1, -50], 0.7, 1),
}


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del_")]

    def tup(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def _mutate(rng, s, err):
    u = rng.random(s.size)
    keep = u >= err * 0.3
    sub = (u >= err * 0.3) & (u < err * 0.5)
    out = s.copy()
    out[sub] = (out[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
    ins = rng.random(s.size) < err * 0.5
    parts = []
    for i in range(s.size):
        if ins[i]:
            parts.append(rng.integers(0, 4))
        if keep[i]:
            parts.append(out[i])
    return np.array(parts, dtype=np.uint8)


def make_pairs(seed, n):
    """(target, read) pairs: two noisy copies of overlapping stretches of an iid genome — dovetails, containments, a few unrelated pairs,
    homopolymer-rich stretches; the read is reverse-complemented for every other pair (the caller of align_hzmaux orients it, wtgbo.c:48-49)"""
    rng = np.random.default_rng(seed)
    G = rng.integers(0, 4, size=60000, dtype=np.uint8)
    for _ in range(200):      # homopolymer runs
        p = int(rng.integers(0, G.size - 20)); G[p:p + int(rng.integers(3, 12))] = G[p]
    pairs = []
    for i in range(n):
        la, lb = int(rng.integers(1500, 9000)), int(rng.integers(1500, 9000))
        a0 = int(rng.integers(0, G.size - la))
        kind = i % 5
        if kind == 4:
            b0 = int(rng.integers(0, G.size - lb))                      # mostly unrelated
        elif kind == 3:
            lb = min(lb, la); b0 = a0 + int(rng.integers(0, la - lb + 1))      # contained
        else:
            b0 = max(0, min(G.size - lb, a0 + int(rng.integers(-lb + 400, la - 400))))
        err = float(rng.choice([0.0, 0.08, 0.15]))
        t = _mutate(rng, G[a0:a0 + la], err) if err else G[a0:a0 + la].copy()
        q = _mutate(rng, G[b0:b0 + lb], err) if err else G[b0:b0 + lb].copy()
        if i % 2:
            q = (3 - q)[::-1].copy()        # a '-' candidate arrives reverse-complemented ... and then never matches strand 0 unless it really is
            if i % 4 == 1:
                t = (3 - t)[::-1].copy()    # ... so half of those get a target of the same orientation
        pairs.append((np.ascontiguousarray(t), np.ascontiguousarray(q)))
    return pairs


@pytest.fixture(scope="module")
def libs():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    return C.CDLL(SHIM), C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


def run_cpu(fn, pairs, prm, min_sm, refine):
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.POINTER(Aln), C.c_void_p, C.c_int]
    p = np.array(prm, dtype=np.int32)
    out = []
    cg = np.zeros(1 << 16, dtype=np.uint32)
    for t, q in pairs:
        a = Aln()
        n = fn(t.ctypes.data, t.size, q.ctypes.data, q.size, p.ctypes.data, min_sm, refine, C.byref(a), cg.ctypes.data, cg.size)
        assert n >= -1
        out.append(None if n < 0 else (a.tup(), cg[:n].copy()))
    return out


def run_device(lib_path, pairs, prm, min_sm, refine):
    from smartdenovo_amd import hipabi
    P = hipabi.Params.defaults(zsize=prm[0], hz=prm[1], kwin=prm[2], ztot=prm[4], zovl=prm[4], max_zmer_freq=prm[5], max_kmer_var=prm[6],
                               w=prm[7], W=prm[8], ew=prm[9], M=prm[11], X=prm[12], O=prm[13], E=prm[15], T=prm[16], min_id=min_sm, refine=refine, aux_strand=1)
    P.kstep = prm[3]
    ctx = hipabi.Context(P, pool_bytes=1 << 30, lib_path=lib_path)
    try:
        seqs = [s for tq in pairs for s in tq]
        ctx.upload(*hipabi.pack_reads(seqs))
        ctx.zindex_build()
        n = len(pairs)
        assert ctx.lib.wtz_batch_begin(ctx.h) == 0
        summ = ctx.pairs_seed(np.arange(n) * 2, np.arange(n) * 2 + 1)
        items = [i for i in range(n) if summ["gate"][i] and summ["nwin"][i][0]]
        out = [None] * n
        if items:
            res, cig = ctx.pairs_align(items, [0] * len(items))
            off = 0
            for k, i in enumerate(items):
                r = res[k]; c = cig[off:off + int(r["cigar_len"])].copy(); off += int(r["cigar_len"])
                if r["n_regs"] == 0:
                    continue
                x = tuple(int(r[f]) for f in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del"))
                if not refine:      # hzm_aln.h:1715-1718, the caller's part of the contract (wtgbo_core.h gbo_hit_passes)
                    tl, ql = pairs[i][0].size, pairs[i][1].size
                    beg = max(0, x[3] - x[1]); end = min(ql, x[4] + tl - x[2])
                    f32 = np.float32
                    if x[0] < 0 or f32(x[6]) < f32(x[5]) * f32(min_sm) or f32(x[6]) < f32(end - beg) * f32(min_sm):
                        continue
                out[i] = (x, c)
        return out
    finally:
        ctx.close()


def same(a, b, what):
    assert len(a) == len(b)
    nhit = 0
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), "%s: pair %d hit / no hit differs (%s vs %s)" % (what, i, x and x[0], y and y[0])
        if x is None:
            continue
        nhit += 1
        assert x[0] == y[0], "%s: pair %d kswx_t %s != %s" % (what, i, x[0], y[0])
        assert np.array_equal(x[1], y[1]), "%s: pair %d CIGAR differs" % (what, i)
    return nhit


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_oracle_align_hzmaux_equals_reference(variant, libs):
    ref, ora = libs
    prm, min_sm, refine = VARIANTS[variant]
    pairs = make_pairs(41, 60)
    r = run_cpu(ref.ref_align_hzmaux, pairs, prm, min_sm, refine)
    o = run_cpu(ora.ora_align_hzmaux_c, pairs, prm, min_sm, refine)
    assert same(r, o, "oracle vs reference") >= 20


@pytest.fixture(scope="module")
def emul_lib():
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return os.path.join(ROOT, "tests", "emul", "libwtz_emul.so")


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_emulated_pair_stages_in_aux_form_equal_reference(variant, libs, emul_lib):
    ref, _ = libs
    prm, min_sm, refine = VARIANTS[variant]
    pairs = make_pairs(43, 40)
    r = run_cpu(ref.ref_align_hzmaux, pairs, prm, min_sm, refine)
    d = run_device(emul_lib, pairs, prm, min_sm, refine)
    assert same(r, d, "emulated device vs reference") >= 12


@pytest.mark.gpu
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_gpu_pair_stages_in_aux_form_equal_reference(variant, libs):
    ref, _ = libs
    prm, min_sm, refine = VARIANTS[variant]
    pairs = make_pairs(47, 300)
    r = run_cpu(ref.ref_align_hzmaux, pairs, prm, min_sm, refine)
    d = run_device(None, pairs, prm, min_sm, refine)
    assert same(r, d, "MI355X vs reference") >= 100
