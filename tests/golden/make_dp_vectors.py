#!/usr/bin/env python3
"""Function-level vectors of the three banded DPs, dumped from the REFERENCE's own routines.

Run in the build container only: needs oracle/_ref/libref_shim.so (= /root/reference's kswx.h / ksw.c compiled where they lie by
oracle/Makefile; the shim exports kswx_extend_align_core kswx.h:234, kswx_extend_align_shift_core kswx.h:101, ksw_global2 ksw.c:503).

Output (committed): tests/golden/dp_vectors.npz - inputs (2-bit code sequences + call parameters) and the reference's outputs
(kswx_t fields, CIGAR words).  Consumers: tests/test_dp_vectors.py (oracle == vectors, CPU) and tests/test_gpu_dp_forms.py
(every DEVICE form of K-sw1 / K-sw2 / K-sw3 == vectors, through the test-only C-ABI entry wtz_test_dp).

Each problem has a class label naming the shape it was built for (band width -> band columns per lane, trace in LDS or in the pool,
rows beyond an envelope, score beyond the packed arg-max key, strand / reverse-complement views, empty sides, early exit ...).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
SCORES = dict(M=2, X=-5, O=-3, E=-1, T=-50)          # wtzmo.c:1547-1551 defaults


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del_")]

    def tup(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def mutate(rng, s, err):
    u = rng.random(s.size)
    keep = u >= err * 0.3
    sub = (u >= err * 0.3) & (u < err * 0.5)
    out = s.copy()
    out[sub] = (out[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
    ins = rng.random(s.size) < err * 0.5
    res = []
    for i in range(s.size):
        if ins[i]:
            res.append(rng.integers(0, 4))
        if keep[i]:
            res.append(out[i])
    return np.array(res, dtype=np.uint8)


def pair(rng, nq, nt, err=0.15, related=True, homopolymers=False):
    n = max(nq, nt) + 64
    if homopolymers:
        a = np.repeat(rng.integers(0, 4, size=n // 3 + 1, dtype=np.uint8), rng.integers(1, 6, size=n // 3 + 1))[:n]
    else:
        a = rng.integers(0, 4, size=n, dtype=np.uint8)
    if related:
        q, t = mutate(rng, a, err), mutate(rng, a, err)
        while q.size < nq or t.size < nt:       # rare: deletions ate too much
            a = np.concatenate([a, rng.integers(0, 4, size=64, dtype=np.uint8)])
            q, t = mutate(rng, a, err), mutate(rng, a, err)
    else:
        q, t = rng.integers(0, 4, size=nq, dtype=np.uint8), rng.integers(0, 4, size=nt, dtype=np.uint8)
    return np.ascontiguousarray(q[:nq]), np.ascontiguousarray(t[:nt])


def call_ext(fn, q, t, init, W):
    a = Aln()
    cg = np.zeros(q.size + t.size + 8, dtype=np.uint32)
    n = fn(int(q.size), C.c_void_p(q.ctypes.data), int(t.size), C.c_void_p(t.ctypes.data), 1, int(init), int(W),
           SCORES["M"], SCORES["X"], SCORES["O"], SCORES["O"], SCORES["E"], SCORES["T"], C.byref(a), C.c_void_p(cg.ctypes.data))
    return a.tup(), cg[:n].copy()


def call_global(ref, q, t, w):
    s = C.c_int()
    cg = np.zeros(q.size + t.size + 8, dtype=np.uint32)
    qq = np.ascontiguousarray(np.concatenate([q, [0]]).astype(np.uint8))
    tt = np.ascontiguousarray(np.concatenate([t, [0]]).astype(np.uint8))
    # hzm_aln.h:1407 passes -I, -E, -D, -E with I = D = O
    n = ref.ref_global(int(q.size), C.c_void_p(qq.ctypes.data), int(t.size), C.c_void_p(tt.ctypes.data), SCORES["M"], SCORES["X"],
                       -SCORES["O"], -SCORES["E"], -SCORES["O"], -SCORES["E"], int(w), C.byref(s), C.c_void_p(cg.ctypes.data))
    cg = cg[:n].copy()
    mat = mis = ins = dele = 0
    x1 = x2 = 0
    for c in cg:
        op, ln = int(c & 0xF), int(c >> 4)
        if op == 0:
            eq = int((q[x1:x1 + ln] == t[x2:x2 + ln]).sum())
            mat += eq
            mis += ln - eq
            x1 += ln
            x2 += ln
        elif op == 1:
            x1 += ln
            ins += ln
        else:
            x2 += ln
            dele += ln
    return (s.value, 0, 0, 0, 0, mat + mis + ins + dele, mat, mis, ins, dele), cg


def main():
    if not os.path.exists(SHIM):
        sys.exit("build the reference shim first: make -C oracle ref")
    ref = C.CDLL(SHIM)
    rng = np.random.default_rng(20260928)
    probs = []      # dict(kind, cls, q, t, init, W, w_param)

    def add(kind, cls, q, t, init=0, W=0, w_param=50):
        probs.append(dict(kind=kind, cls=cls, q=q, t=t, init=int(init), W=int(W), w_param=int(w_param)))

    # ---------------- K-sw3: kswx_extend_align_shift_core (W negative = exact band, hzm_aln.h:1361-1374) ----------------
    for W, cls, n in ((30, "s_c1", 6), (100, "s_c4", 8), (250, "s_c8", 5), (380, "s_c12", 3), (500, "s_c16", 3), (630, "s_c20", 2), (760, "s_c24", 2),
                      (800, "s_c28", 4), (1000, "s_c32", 2)):
        for k in range(n):
            L = int(rng.integers(2 * W + 40, 2 * W + 700))
            q, t = pair(rng, L + int(rng.integers(-30, 30)), L + int(rng.integers(-30, 30)), err=float(rng.choice([0.1, 0.15, 0.2])))
            add(0, cls, q, t, init=int(rng.integers(0, 500)), W=-W)
    for k in range(10):      # shorter than the band: n_col = tl, W clipped to max(qlen, tlen)
        nq, nt = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        q, t = pair(rng, nq, nt)
        add(0, "s_short", q, t, init=int(rng.integers(0, 300)), W=-800)
    for ql in (63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1025):      # 64-row trace chunks, the four-wave threshold
        q, t = pair(rng, ql, ql + int(rng.integers(-20, 20)), err=0.12)
        add(0, "s_rows", q, t, init=200, W=-800)
    for k in range(6):       # unrelated sequences: early exit when a row maximum is <= 0 (kswx.h:185)
        q, t = pair(rng, int(rng.integers(300, 3000)), int(rng.integers(300, 3000)), related=False)
        add(0, "s_stop", q, t, init=int(rng.integers(0, 120)), W=-800)
    for k in range(6):       # one side ends first: the gmax / T end rule (kswx.h:200-206, 216-217)
        nq = int(rng.integers(200, 1500))
        q, t = pair(rng, nq, nq + int(rng.choice([-150, -60, 60, 150])), err=0.15)
        add(0, "s_end", q, t, init=int(rng.integers(0, 400)), W=-800)
    for k in range(4):       # homopolymer runs: ties everywhere
        q, t = pair(rng, 900, 900, err=0.15, homopolymers=True)
        add(0, "s_homo", q, t, init=100, W=-int(rng.choice([100, 800])))
    q, t = pair(rng, 2400, 2400, err=0.15)
    add(0, "s_wide", q, t, init=300, W=-1100)         # 2201 columns: beyond 64 lanes x 32 and the LDS rings -> scalar fallback inside the job kernel
    q, t = pair(rng, 700, 700, err=0.15)
    add(0, "s_keyovf", q, t, init=(1 << 20) - 300, W=-100)     # init + M*min(ql,tl) >= 2^20: packed arg-max key overflows -> general kernel
    q, t = pair(rng, 33100, 33100, err=0.15)
    add(0, "s_longt", q, t, init=100, W=-30)          # target beyond the 1032 LDS words of the register kernels
    add(0, "s_empty", np.zeros(0, np.uint8), pair(rng, 50, 50)[1], init=77, W=-800)
    add(0, "s_empty", pair(rng, 50, 50)[0], np.zeros(0, np.uint8), init=0, W=-800)
    add(0, "s_neginit", *pair(rng, 300, 300), init=-40, W=-800)

    # ---------------- K-sw1: kswx_extend_align_core (band = -w, clipped by max_gap; hzm_aln.h:1268-1272) ----------------
    for w, cls, n in ((50, "f_w50", 14), (20, "f_w20", 6), (5, "f_w5", 4), (100, "f_w100", 6), (200, "f_w200", 6)):
        for k in range(n):
            nq = int(rng.integers(1, 4 * w + 60)) if k % 3 else int(rng.integers(1, 70))
            nt = max(1, nq + int(rng.integers(-w // 2 - 5, w // 2 + 5)))
            q, t = pair(rng, nq, nt, err=float(rng.choice([0.05, 0.15, 0.3])))
            add(1, cls, q, t, init=int(rng.integers(0, 2000)), w_param=w)
    for w in (50, 100):      # long problems: the 4-bit trace leaves the LDS slice
        for k in range(3):
            nq = int(rng.integers(500, 2040))
            q, t = pair(rng, nq, nq + int(rng.integers(-40, 40)))
            add(1, "f_long", q, t, init=int(rng.integers(0, 500)), w_param=w)
    q, t = pair(rng, 2300, 2310)
    add(1, "f_rows2048", q, t, init=50, w_param=50)            # more than 2048 rows: scalar body
    q, t = pair(rng, 80, 80)
    add(1, "f_keyovf", q, t, init=(1 << 23) - 50, w_param=50)  # |h| beyond the packed key: scalar body
    for k in range(4):
        q, t = pair(rng, int(rng.integers(20, 200)), int(rng.integers(20, 200)), related=False)
        add(1, "f_stop", q, t, init=int(rng.integers(0, 60)), w_param=50)
    add(1, "f_empty", np.zeros(0, np.uint8), pair(rng, 30, 30)[1], init=120, w_param=50)
    add(1, "f_empty", pair(rng, 30, 30)[0], np.zeros(0, np.uint8), init=0, w_param=50)
    for k in range(3):
        q, t = pair(rng, 120, 120, homopolymers=True)
        add(1, "f_homo", q, t, init=300, w_param=50)

    # ---------------- K-sw2: ksw_global2 (one call at band w; the band-doubling loop stays with the caller, hzm_aln.h:1400-1417) ----------------
    for w, cls, n in ((25, "g_c1", 6), (50, "g_c2", 10), (100, "g_c4", 5), (200, "g_c8", 5), (255, "g_c8", 2), (400, "g_ring", 3), (1600, "g_ringwide", 2)):
        for k in range(n):
            nq = int(rng.integers(max(2, w // 2), 2 * w + 500))
            nt = max(1, nq + int(rng.integers(-w, w + 1)))
            q, t = pair(rng, nq, nt, err=float(rng.choice([0.1, 0.15, 0.25])))
            add(2, cls, q, t, W=w)
    for k in range(8):
        nq, nt = int(rng.integers(1, 64)), int(rng.integers(1, 64))
        w = 50
        while w < abs(nq - nt):
            w <<= 1
        q, t = pair(rng, nq, nt)
        add(2, "g_small", q, t, W=w)
    for nt in (2047, 2048, 2049, 2500, 5000):       # gaps longer than 2048 rows: pool trace, target words reloaded per 2048-row block
        q, t = pair(rng, nt + int(rng.integers(-40, 40)), nt)
        add(2, "g_long", q, t, W=50)
    for k in range(3):
        q, t = pair(rng, int(rng.integers(100, 400)), int(rng.integers(100, 400)), related=False)
        w = 50
        while w < abs(q.size - t.size):
            w <<= 1
        add(2, "g_unrelated", q, t, W=w)
    add(2, "g_empty", np.zeros(0, np.uint8), pair(rng, 40, 40)[1], W=50)       # ksw.c:511-524, 571-579
    add(2, "g_empty", pair(rng, 40, 40)[0], np.zeros(0, np.uint8), W=50)
    add(2, "g_empty", np.zeros(0, np.uint8), np.zeros(0, np.uint8), W=50)
    for k in range(3):
        q, t = pair(rng, 300, 310, homopolymers=True)
        add(2, "g_homo", q, t, W=50)

    # ---------------- expected outputs from the reference ----------------
    alns, cigs = [], []
    for p in probs:
        if p["kind"] == 0:
            a, cg = call_ext(ref.ref_extend_shift, p["q"], p["t"], p["init"], p["W"])
        elif p["kind"] == 1:
            a, cg = call_ext(ref.ref_extend_fixed, p["q"], p["t"], p["init"], p["w_param"])
        else:
            a, cg = call_global(ref, p["q"], p["t"], p["W"])
        alns.append(a)
        cigs.append(cg)
    # ---------------- views: every (reverse-complement view, walking direction) combination the path uses ----------------
    seqs, view = [], []
    for i, p in enumerate(probs):
        row = []
        for side in ("q", "t"):
            L = p[side]
            rev = int(rng.integers(0, 2))
            strand = int(rng.choice([1, -1]))
            fa, fb = int(rng.integers(0, 40)), int(rng.integers(0, 40))
            seg = L if strand > 0 else L[::-1]
            v = np.concatenate([rng.integers(0, 4, size=fa, dtype=np.uint8), seg, rng.integers(0, 4, size=fb, dtype=np.uint8)]).astype(np.uint8)
            frm = fa if strand > 0 else fa + L.size - 1
            if L.size == 0:
                frm = min(fa, max(0, v.size - 1))
            if v.size == 0:
                v = np.zeros(1, np.uint8)
                frm = 0
            read = v if not rev else (3 - v)[::-1]
            seqs.append(np.ascontiguousarray(read, dtype=np.uint8))
            row += [rev, frm, strand]
        view.append(row)
    meta = dict(scores=SCORES, classes=sorted(set(p["cls"] for p in probs)),
                note="reads 2i / 2i+1 hold the query / target of problem i; view = (q_rev, q_from, q_strand, t_rev, t_from, t_strand)")
    np.savez_compressed(
        os.path.join(HERE, "dp_vectors.npz"),
        kind=np.array([p["kind"] for p in probs], dtype=np.int32),
        cls=np.array([p["cls"] for p in probs]),
        init=np.array([p["init"] for p in probs], dtype=np.int32),
        W=np.array([p["W"] for p in probs], dtype=np.int32),
        w_param=np.array([p["w_param"] for p in probs], dtype=np.int32),
        qlen=np.array([p["q"].size for p in probs], dtype=np.int32),
        tlen=np.array([p["t"].size for p in probs], dtype=np.int32),
        view=np.array(view, dtype=np.int32),
        read_off=np.cumsum([0] + [s.size for s in seqs]).astype(np.int64),
        read_codes=np.concatenate(seqs).astype(np.uint8),
        aln=np.array(alns, dtype=np.int32),
        cig_off=np.cumsum([0] + [c.size for c in cigs]).astype(np.int64),
        cig=np.concatenate(cigs).astype(np.uint32) if cigs else np.zeros(0, np.uint32),
        meta=np.array(json.dumps(meta)),
    )
    print("%d problems: %s" % (len(probs), {k: int(sum(1 for p in probs if p["kind"] == k)) for k in (0, 1, 2)}))
    print("size %.1f KB" % (os.path.getsize(os.path.join(HERE, "dp_vectors.npz")) / 1024))


if __name__ == "__main__":
    main()
