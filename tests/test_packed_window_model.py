"""The arithmetic of the packed 16-bit K-sw3 form (smartdenovo_amd/csrc/wtz_sw_frame16.h) as a scalar model, against the oracle's
kswx_extend_align_shift_core (kswx.h:101-232) on seeded problems.  Runs without a GPU: what it pins is the ARGUMENT the kernel rests on -
  (1) the anti-diagonal frame G = H - (i+j)E with the row body of wtz_pk_row (m~ = diag + s - 2E, E~' = max(E~, m~ + O), F~' = max(F~, m~ + O),
      four decisions as signs of differences, -10000 slots beyond the band end) gives the reference's result in plain integers;
  (2) window (a): the same with every stored value held as value - bias in a saturating 16-bit range;
  (3) window (b): with an init_score that alone breaks 16 bits, the -10000 family raised to NG (wtz_pk_window) and sums saturating at the bottom
      still give the reference's score, end cell and CIGAR, and the two tests against zero see a family value as negative.
The device kernel itself is compared with the reference's vectors and goldens by the GPU tests (DP form 7 in tests/test_gpu_dp_forms.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

M, X, O, E, T = 2, -5, -3, -1, 1000          # the scores every caller of the path uses (wtzmo.c defaults; T as hzm_aln.h passes it)


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del_")]

    def tup(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


@pytest.fixture(scope="module")
def ora():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


def oracle_shift(ora, q, t, init, W):
    cg = np.zeros(q.size + t.size + 8, dtype=np.uint32)
    a = Aln()
    m = ora.ora_extend_shift_c(int(q.size), C.c_void_p(q.ctypes.data), int(t.size), C.c_void_p(t.ctypes.data), 1, int(init), int(W), M, X, O, O, E, T, C.byref(a), C.c_void_p(cg.ctypes.data))
    return a.tup(), cg[:m].tolist()


def geometry(qlen, tlen, W):
    """wtz_ext_geometry for a fixed band (W < 0: the width itself, as hzm_aln.h passes it)"""
    w = min(-W, max(qlen, tlen))
    if qlen < tlen:
        ql, tl = qlen, (qlen + w if qlen + w < tlen else tlen)
    else:
        tl, ql = tlen, (tlen + w if tlen + w < qlen else qlen)
    return w, ql, tl


def pk_window(init, ql, tl):
    """wtz_pk_window: (bias, NG, SH) or None"""
    aE, aO, aX = -E, -O, -X
    Xp = X - 2 * E
    Xm = min(Xp, 0)
    mn = min(ql, tl)
    hi = init + M * mn + (ql + tl + 4) * aE + (M - X) + 64
    NG = -10000
    lo = NG + (ql + 3) * Xm + O - 33 * aE - 64
    SH = -(1 << 30)
    if hi - lo > 65000:
        if M * mn >= 10000:
            return None
        step = max(aX, aO + aE)
        rlow = init - aO - (tl + 1) * aE - (ql + 1) * step - (ql + 2) * aO
        NG = rlow - M * mn - 64
        if NG <= -10000:
            return None
        lo = NG - 33 * aE - 64
        if hi - lo > 65000:
            return None
        SH = NG + M * mn
    return (hi + lo) // 2, NG, SH


def model_shift(q, t, init, W, window=None):
    """The row loop of wtz_extend_shift_pk on one 'lane' that owns every column.  window = None: plain integers, the reference's -10000;
    window = (bias, NG, SH): values kept as value - bias, the three sums of the row that can fall saturate at the bottom of 16 bits."""
    qlen, tlen = q.size, t.size
    init = max(int(init), 0)
    w, ql, tl = geometry(qlen, tlen, W)
    bias, NG, SH = (0, -10000, -(1 << 30)) if window is None else window
    if window is None:
        sat = lambda v: v
    else:
        sat = lambda v: max(-32768, min(32767, v))
    Xp = X - 2 * E
    # row -1 in the frame: G(-1, c) = H(-1, c) - (c - 1)E, E~(0, c) = NG - cE; column -1 is index 0
    H = [0] * (tl + 3)
    Ev = [0] * (tl + 3)
    for c in range(-1, tl + 1):
        hr = init if c < 0 else init + O + E * (c + 1)
        H[c + 1] = hr - (c - 1) * E - bias
        Ev[c + 1] = NG - c * E - bias
    zb, trace = [], []
    mx, mi, mj, gmax, gi, gj = init, -1, -1, 0, -1, -1
    c_ = 0
    jb, je = 0, min(tl, w + 1)
    jbp, jep = -1, tl + 1            # row -1 "band": columns -1 .. tl
    rows = 0
    for i in range(ql):
        rows += 1
        nH, nE = list(H), list(Ev)
        f = NG - (i + jb) * E - bias
        bnd = ((init + O + E * i) if jb == 0 else NG) - (i + jb - 2) * E - bias
        best_v, best_j = None, -1
        row = []
        for j in range(jb, je):
            src = H[j] if j - 1 >= jbp else bnd                       # H[j] is column j - 1
            e = Ev[j + 1]
            s = M if q[i] == t[j] else X
            m = sat(src + (s - X) + Xp) if window is not None else src + s - 2 * E
            h0 = max(m, e)
            d = (8 if m < e else 0) | (4 if h0 < f else 0)
            h = max(h0, f)
            tt = sat(m + O)
            d |= (2 if tt < e else 0) | (1 if tt < f else 0)
            en = max(e, tt)
            f = max(f, tt)
            nH[j + 1] = h
            nE[j + 1] = en
            row.append(d)
            v = sat(h + (j - jb) * E)                                 # the row maximum's value: h + column * E (first arg-max)
            if best_v is None or v > best_v:
                best_v, best_j = v, j
        zb.append(jb)
        trace.append(row)
        Hm = best_v + bias + (i + jb) * E
        imax, mj2 = 0, -1
        if Hm > 0 and Hm > SH:
            imax, mj2 = Hm, best_j
        H, Ev = nH, nE
        if je == tlen:
            h1 = H[je] + bias + (i + je - 1) * E
            if h1 > SH and gmax < h1:
                gmax, gi, gj = h1, i, je - 1
        if i + 1 == qlen and gmax < imax:
            gmax, gi, gj = imax, i, mj2
        jbp, jep = jb, je
        stop = False
        if imax > mx:
            mx, mi, mj = imax, i, mj2
        elif imax <= 0:
            stop = True
        if stop:
            break
        c_ += 1
        if c_ < mj2:
            c_ += 1
        elif c_ > mj2:
            c_ -= 1
        jb, je = max(0, c_ - w), min(tl, c_ + w + 1)
        # the slots the next row reads beyond this row's band end
        H[jep + 1] = NG - (i + jep) * E - bias
        Ev[jep + 1] = NG - (i + 1 + jep) * E - bias
        if jep + 2 < len(Ev):
            Ev[jep + 2] = NG - (i + 2 + jep) * E - bias
    if gmax > 0 and gmax >= mx + T:
        score, qe, te = gmax, gi, gj
    else:
        score, qe, te = mx, mi, mj
    # the walk of wtz_shift_traceback
    i_, j_, st = qe, te, 0
    ops = []
    mat = mis = ins = dele = 0
    while i_ >= 0 and j_ >= 0:
        cc = j_ - zb[i_]
        d = trace[i_][cc] if 0 <= cc < len(trace[i_]) else 0
        if st == 0:
            st = 2 if (d & 4) else (1 if (d & 8) else 0)
        elif st == 1:
            st = 1 if (d & 2) else 0
        else:
            st = 2 if (d & 1) else 0
        if st == 0:
            if q[i_] == t[j_]:
                mat += 1
            else:
                mis += 1
            i_ -= 1; j_ -= 1
        elif st == 1:
            i_ -= 1; ins += 1
        else:
            j_ -= 1; dele += 1
        ops.append(st)
    if i_ >= 0:
        ins += i_ + 1; ops += [1] * (i_ + 1)
    if j_ >= 0:
        dele += j_ + 1; ops += [2] * (j_ + 1)
    ops.reverse()
    cigar = []
    for op in ops:
        if cigar and (cigar[-1] & 15) == op:
            cigar[-1] += 16
        else:
            cigar.append(16 | op)
    return (score, 0, te + 1, 0, qe + 1, mat + mis + ins + dele, mat, mis, ins, dele), cigar, rows


def mutate(rng, s, err):
    out = []
    for b in s:
        u = rng.random()
        if u < err * 0.5:
            out.append(int(rng.integers(0, 4))); out.append(int(b))
        elif u < err * 0.8:
            continue
        elif u < err:
            out.append(int((b + 1 + rng.integers(0, 3)) % 4))
        else:
            out.append(int(b))
    return np.array(out, dtype=np.uint8)


def problems():
    rng = np.random.Generator(np.random.PCG64(61))
    out = []
    for k in range(36):
        L = int(rng.integers(40, 260))
        seg = rng.integers(0, 4, size=L, dtype=np.uint8)
        kind = k % 4
        if kind == 3:            # unrelated sequences: every real path loses on every row
            q, t = rng.integers(0, 4, size=L, dtype=np.uint8), rng.integers(0, 4, size=int(L * 1.1), dtype=np.uint8)
        elif kind == 2:          # homologous start, unrelated tail: the alignment runs on through the tail on its init score
            q = np.concatenate([mutate(rng, seg[: L // 2], 0.15), rng.integers(0, 4, size=L // 2, dtype=np.uint8)])
            t = np.concatenate([mutate(rng, seg[: L // 2], 0.15), rng.integers(0, 4, size=L // 2 + 20, dtype=np.uint8)])
        else:
            q, t = mutate(rng, seg, 0.15), mutate(rng, seg, 0.15)
        W = -int(rng.choice([8, 20, 50, 400]))
        for init in (0, 700, 24000, 47000, (1 << 20) - 300):
            out.append((np.ascontiguousarray(q), np.ascontiguousarray(t), init, W))
    return out


def test_frame_model_and_both_windows_equal_the_oracle(ora):
    n_a = n_b = 0
    for q, t, init, W in problems():
        want = oracle_shift(ora, q, t, init, W)
        got = model_shift(q, t, init, W)
        assert (got[0], got[1]) == want, "plain frame model: init %d W %d ql %d tl %d" % (init, W, q.size, t.size)
        _, ql, tl = geometry(q.size, t.size, W)
        win = pk_window(max(init, 0), ql, tl)
        assert win is not None
        got16 = model_shift(q, t, init, W, win)
        assert (got16[0], got16[1]) == want, "16-bit window %s: init %d W %d ql %d tl %d" % (win, init, W, q.size, t.size)
        if win[1] == -10000:
            n_a += 1
        else:
            n_b += 1
            assert win[1] > -10000 and win[2] == win[1] + M * min(ql, tl)
    assert n_a > 30 and n_b > 30          # both windows exercised


def test_raised_family_is_seen_as_negative_by_the_tests_against_zero(ora):
    """a band that leaves the real values behind: the row maximum is a family value (positive as a number in window (b)) and must end the job like the reference's -10000 + x does"""
    rng = np.random.Generator(np.random.PCG64(5))
    hits = 0
    for _ in range(40):
        L = int(rng.integers(30, 90))
        q = rng.integers(0, 4, size=L, dtype=np.uint8)
        t = rng.integers(0, 4, size=L + 40, dtype=np.uint8)
        for init in (70000, 900000):
            want = oracle_shift(ora, q, t, init, -3)
            _, ql, tl = geometry(q.size, t.size, -3)
            win = pk_window(init, ql, tl)
            assert win is not None and win[1] > 0          # the family stands above zero as a number
            got = model_shift(q, t, init, -3, win)
            assert (got[0], got[1]) == want
            hits += 1
    assert hits == 80
