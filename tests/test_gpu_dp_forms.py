"""Every DEVICE form of the three banded DPs against function-level vectors dumped from the reference's own routines
(tests/golden/dp_vectors.npz <- kswx_extend_align_shift_core kswx.h:101, kswx_extend_align_core kswx.h:234, ksw_global2 ksw.c:503),
driven through the test-only C-ABI entry wtz_test_dp (include/wtzmo_hip.h).  The forms are picked by the same functions the product
kernels call; form 0 is the product's own choice.  A form may decline a problem outside its envelope (form_used == 0), but what it
answers must equal the reference bit for bit, and every shape class must be answered by the forms listed for it."""
import numpy as np
import pytest

from dpvec import Vectors
from smartdenovo_amd import hipabi

pytestmark = pytest.mark.gpu
FIELDS = ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del")


@pytest.fixture(scope="module")
def vec():
    return Vectors()


def make_ctx(vec, **kw):
    ctx = hipabi.Context(hipabi.Params.defaults(**kw), pool_bytes=2 << 30)
    ctx.upload(*hipabi.pack_reads(vec.reads()))
    return ctx


def problems(vec, idx):
    p = np.zeros(len(idx), dtype=hipabi.DP_PROBLEM)
    for k, i in enumerate(idx):
        v = vec.view[i]
        p[k] = (2 * i, 2 * i + 1, v[0], v[3], v[1], v[4], v[2], v[5], vec.qlen[i], vec.tlen[i], vec.init[i], vec.W[i])
    return p


def check(vec, idx, out, cigs, kind, form, must):
    """out == reference wherever the form answered; `must`(class) -> the form has to answer problems of that class"""
    answered = {}
    for k, i in enumerate(idx):
        cls = str(vec.cls[i])
        used = int(out["form_used"][k])
        answered.setdefault(cls, [0, 0])
        answered[cls][1] += 1
        if used == 0:
            assert not must(cls), "kind %d form %d declined problem %d of class %s" % (kind, form, i, cls)
            continue
        answered[cls][0] += 1
        exp = vec.aln[i]
        got = tuple(int(out[f][k]) for f in FIELDS)
        if kind == 2:       # ksw_global2 returns the score; the counts are the fold the gap task makes over the CIGAR
            exp_t = (int(exp[0]), int(exp[5]), int(exp[6]), int(exp[7]), int(exp[8]), int(exp[9]))
            got_t = (got[0], got[5], got[6], got[7], got[8], got[9])
        else:
            exp_t, got_t = tuple(int(x) for x in exp), got
        assert got_t == exp_t, "kind %d form %d (used %d) problem %d class %s: %s != %s" % (kind, form, used, i, cls, got_t, exp_t)
        ec = vec.expected_cigar(i)
        assert cigs[k].size == ec.size and (cigs[k] == ec).all(), "kind %d form %d (used %d) problem %d class %s: CIGAR differs" % (kind, form, used, i, cls)
    return answered


# K-sw3: which forms must answer which class.  1 = one-wave register kernel (band <= 64 x 32 columns, target <= 1032 LDS words, key range),
# (2 = the round-4 four-wave kernel: retired in round 6), 3 = LDS-ring kernel incl. its scalar fallback (everything), 4 = scalar body (everything),
# 5 = one-wave kernel in the anti-diagonal frame (round 5, the product's one-wave form: same envelope as 1),
# 6 = the frame form on four wavefronts (round 6; band <= 256 x 8), 7 = the frame form with two 16-bit cells per register (round 6: the product's first choice; same band envelope, values inside a 16-bit window)
SHIFT_MUST = {
    1: {"s_c1", "s_c4", "s_c8", "s_c12", "s_c16", "s_c20", "s_c24", "s_c28", "s_c32", "s_short", "s_rows", "s_stop", "s_end", "s_homo", "s_neginit"},
    5: {"s_c1", "s_c4", "s_c8", "s_c12", "s_c16", "s_c20", "s_c24", "s_c28", "s_c32", "s_short", "s_rows", "s_stop", "s_end", "s_homo", "s_neginit"},
    6: {"s_c1", "s_c4", "s_c8", "s_c12", "s_c16", "s_c20", "s_c24", "s_c28", "s_c32", "s_short", "s_rows", "s_stop", "s_end", "s_homo", "s_neginit"},
    7: {"s_c1", "s_c4", "s_c8", "s_c12", "s_c16", "s_c20", "s_c24", "s_c28", "s_c32", "s_short", "s_rows", "s_stop", "s_end", "s_homo", "s_neginit"},
}


@pytest.mark.parametrize("form", [0, 1, 3, 4, 5, 6, 7])
def test_shift_extension_forms(form, vec):
    idx = [int(i) for i in np.nonzero(vec.kind == 0)[0]]
    ctx = make_ctx(vec)
    try:
        out, cigs = ctx.test_dp(hipabi.DP_SHIFT, form, problems(vec, idx))
    finally:
        ctx.close()
    must = (lambda c: True) if form in (0, 3, 4) else (lambda c: c in SHIFT_MUST[form])
    ans = check(vec, idx, out, cigs, 0, form, must)
    if form == 0:       # the product's dispatch: the register kernels take what they can, the general kernel the rest
        used = {str(vec.cls[i]): int(out["form_used"][k]) for k, i in enumerate(idx)}
        assert used["s_wide"] == 3 and used["s_keyovf"] in (3, 7) and used["s_longt"] == 3 and used["s_c4"] in (1, 5, 6, 7)
    if form in (1, 5, 6, 7):  # outside the register kernels' envelope
        # (the packed form keeps values relative to a per-job bias: an init_score near 2^20 is inside its window, and its answer is checked like any other)
        for cls in ("s_wide", "s_longt", "s_empty") + (() if form == 7 else ("s_keyovf",)):
            assert ans[cls][0] == 0, "form %d should decline %s" % (form, cls)


def _fixed_must(form, w):
    # the product's choice and the scalar body answer everything; a forced register form may decline what is outside its envelope
    # (coverage of the forced forms is asserted through the totals in the test)
    return (lambda cls: True) if form in (0, 255) else (lambda cls: False)


@pytest.mark.parametrize("w", [50, 20, 5, 100, 200])
def test_fixed_extension_forms(w, vec):
    idx = [int(i) for i in np.nonzero((vec.kind == 1) & (vec.w_param == w))[0]]
    assert idx
    ctx = make_ctx(vec, w=w)
    total = {}
    try:
        for form in (0, 1, 2, 17, 18, 20, 24, 64, 255):
            out, cigs = ctx.test_dp(hipabi.DP_FIXED, form, problems(vec, idx))
            ans = check(vec, idx, out, cigs, 1, form, _fixed_must(form, w))
            total[form] = sum(a for a, _ in ans.values())
            if form == 0:
                used = {int(u) for u in out["form_used"]}
                total["auto_forms"] = used
    finally:
        ctx.close()
    assert total[0] == len(idx) and total[255] == len(idx)
    if w == 50:     # the path's default band: 1 / 2 columns per lane in LDS, the long problems in the pool, rows > 2048 and the key overflow scalar
        assert {1, 2, 18, 255} <= total["auto_forms"] and total[1] > 0 and total[2] > 0 and total[17] > 0 and total[18] > 0
    if w in (50, 20, 5):      # form 64 = one lane per problem (wtz_sw_lane.h, absolute mode): everything with a band of <= 104 columns and <= 511 rows
        assert total[64] > 0.5 * len(idx), "the lane form answered only %d of %d problems" % (total[64], len(idx))
    if w == 100:
        assert total[20] > 0 and 20 in total["auto_forms"]
    if w == 200:
        assert total[24] > 0 and 24 in total["auto_forms"]


def test_global_forms(vec):
    idx = [int(i) for i in np.nonzero(vec.kind == 2)[0]]
    ctx = make_ctx(vec)
    total = {}
    try:
        for form in (0, 1, 2, 17, 18, 20, 24, 32, 33, 64, 255):
            out, cigs = ctx.test_dp(hipabi.DP_GLOBAL, form, problems(vec, idx))
            ans = check(vec, idx, out, cigs, 2, form, (lambda c: True) if form in (0, 255) else (lambda c: False))
            total[form] = sum(a for a, _ in ans.values())
            if form == 0:
                used = {str(vec.cls[i]): set() for i in idx}
                for k, i in enumerate(idx):
                    used[str(vec.cls[i])].add(int(out["form_used"][k]))
                # the product's choice per class: LDS trace for short narrow gaps, pool trace for long / wide ones, the scalar body for empty sides
                assert used["g_c4"] <= {18, 20} and 20 in used["g_c4"] and used["g_c8"] <= {20, 24} and 24 in used["g_c8"] and used["g_long"] <= {17, 18} and 255 in used["g_empty"]
            if form == 33:
                cls33 = {str(vec.cls[i]) for k, i in enumerate(idx) if out["form_used"][k] == 33}
                assert {"g_ring", "g_ringwide"} <= cls33
    finally:
        ctx.close()
    assert total[0] == len(idx) and total[255] == len(idx)
    for form in (1, 2, 17, 18, 20, 24, 32, 33, 64):      # 64 = one lane per gap (wtz_lane_global)
        assert total[form] > 0, "form %d answered nothing" % form
