"""The oracle's three banded DPs against the committed function-level vectors of the reference (tests/golden/dp_vectors.npz:
kswx_extend_align_shift_core kswx.h:101, kswx_extend_align_core kswx.h:234, ksw_global2 ksw.c:503).  Runs anywhere (no reference, no
GPU): this is what pins the oracle's DPs on the GPU box, where /root/reference does not exist."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dpvec import Vectors


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del_")]

    def tup(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


@pytest.fixture(scope="module")
def ora():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


@pytest.fixture(scope="module")
def vec():
    return Vectors()


def test_vectors_cover_every_shape_class(vec):
    want = {"s_c1", "s_c4", "s_c8", "s_c12", "s_c16", "s_c20", "s_c24", "s_c28", "s_c32", "s_short", "s_rows", "s_stop", "s_end", "s_homo", "s_wide",
            "s_keyovf", "s_longt", "s_empty", "s_neginit", "f_w50", "f_w20", "f_w5", "f_w100", "f_w200", "f_long", "f_rows2048", "f_keyovf", "f_stop",
            "f_empty", "f_homo", "g_c1", "g_c2", "g_c4", "g_c8", "g_ring", "g_ringwide", "g_small", "g_long", "g_unrelated", "g_empty", "g_homo"}
    assert want <= set(vec.cls.tolist())
    # the views cover both strands and both walking directions on both sides
    assert {tuple(v) for v in vec.view[:, [0, 2]].tolist()} == {(0, 1), (0, -1), (1, 1), (1, -1)}
    assert {tuple(v) for v in vec.view[:, [3, 5]].tolist()} == {(0, 1), (0, -1), (1, 1), (1, -1)}


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_oracle_equals_reference_vectors(kind, vec, ora):
    S = vec.meta["scores"]
    n = 0
    for i in np.nonzero(vec.kind == kind)[0]:
        q, t = vec.logical(i, "q"), vec.logical(i, "t")
        cg = np.zeros(q.size + t.size + 8, dtype=np.uint32)
        if kind == 2:
            s = C.c_int()
            qq = np.ascontiguousarray(np.concatenate([q, [0]]).astype(np.uint8))
            tt = np.ascontiguousarray(np.concatenate([t, [0]]).astype(np.uint8))
            m = ora.ora_global_c(int(q.size), C.c_void_p(qq.ctypes.data), int(t.size), C.c_void_p(tt.ctypes.data), S["M"], S["X"], -S["O"], -S["E"], -S["O"], -S["E"],
                                 int(vec.W[i]), C.byref(s), C.c_void_p(cg.ctypes.data))
            assert s.value == vec.aln[i][0], "problem %d (%s)" % (i, vec.cls[i])
        else:
            a = Aln()
            fn = ora.ora_extend_shift_c if kind == 0 else ora.ora_extend_fixed_c
            W = int(vec.W[i]) if kind == 0 else int(vec.w_param[i])
            m = fn(int(q.size), C.c_void_p(q.ctypes.data), int(t.size), C.c_void_p(t.ctypes.data), 1, int(vec.init[i]), W, S["M"], S["X"], S["O"], S["O"], S["E"], S["T"],
                   C.byref(a), C.c_void_p(cg.ctypes.data))
            assert a.tup() == tuple(vec.aln[i].tolist()), "problem %d (%s)" % (i, vec.cls[i])
        assert (cg[:m] == vec.expected_cigar(i)).all() and m == vec.expected_cigar(i).size, "problem %d (%s): CIGAR" % (i, vec.cls[i])
        n += 1
    assert n > 40
