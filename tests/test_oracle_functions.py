"""Function-level pinning of the oracle against the REAL reference routines exported by oracle/_ref/libref_shim.so
(built from /root/reference by `make -C oracle ref`): the three banded DPs incl. CIGARs, the unstable sort on
tie-heavy keys, the quick-select median."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SHIM), reason="reference shim not built (needs /root/reference once)")


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del_")]

    def tup(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


@pytest.fixture(scope="module")
def libs():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    return C.CDLL(SHIM), C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


def _mutate(rng, s, err):
    out = []
    for b in s:
        u = rng.random()
        if u < err * 0.3:
            continue
        if u < err * 0.5:
            b = (b + rng.integers(1, 4)) & 3
        if rng.random() < err * 0.5:
            out.append(rng.integers(0, 4))
        out.append(b)
    return np.array(out, dtype=np.uint8)


def _pair(rng, n, err=0.15):
    a = rng.integers(0, 4, size=n, dtype=np.uint8)
    return np.ascontiguousarray(_mutate(rng, a, err)), np.ascontiguousarray(_mutate(rng, a, err))


@pytest.mark.parametrize("kind", ["fixed", "shift"])
def test_extension_dps(kind, libs):
    ref, ora = libs
    rng = np.random.default_rng(5)
    fr, fo = (ref.ref_extend_fixed, ora.ora_extend_fixed_c) if kind == "fixed" else (ref.ref_extend_shift, ora.ora_extend_shift_c)
    for it in range(300):
        n = int(rng.integers(1, 60 if kind == "fixed" else 900))
        q, t = _pair(rng, n, err=float(rng.choice([0.05, 0.15, 0.3])))
        if q.size == 0 or t.size == 0:
            continue
        strand = 1 if kind == "fixed" or it % 2 == 0 else -1
        W = int(rng.choice([50, 20, 5])) if kind == "fixed" else -int(rng.choice([800, 100, 30]))
        init = int(rng.integers(0, 400))
        qp = q.ctypes.data + (q.size - 1 if strand < 0 else 0)
        tp = t.ctypes.data + (t.size - 1 if strand < 0 else 0)
        a, b = Aln(), Aln()
        ca = np.zeros(q.size + t.size + 8, dtype=np.uint32)
        cb = np.zeros_like(ca)
        na = fr(int(q.size), C.c_void_p(qp), int(t.size), C.c_void_p(tp), strand, init, W, 2, -5, -3, -3, -1, -50, C.byref(a), C.c_void_p(ca.ctypes.data))
        nb = fo(int(q.size), C.c_void_p(qp), int(t.size), C.c_void_p(tp), strand, init, W, 2, -5, -3, -3, -1, -50, C.byref(b), C.c_void_p(cb.ctypes.data))
        assert a.tup() == b.tup() and na == nb and (ca[:na] == cb[:nb]).all()


def test_global_dp_including_empty_sides(libs):
    ref, ora = libs
    rng = np.random.default_rng(6)
    for it in range(300):
        n = int(rng.integers(0, 200))
        q, t = _pair(rng, n) if n else (np.zeros(0, np.uint8), np.zeros(0, np.uint8))
        if it % 10 == 0:
            q = np.zeros(0, np.uint8)
        if it % 13 == 0:
            t = np.zeros(0, np.uint8)
        w = 50
        while w < abs(int(q.size) - int(t.size)):
            w <<= 1
        sa, sb = C.c_int(), C.c_int()
        ca = np.zeros(q.size + t.size + 8, dtype=np.uint32)
        cb = np.zeros_like(ca)
        qq = np.ascontiguousarray(np.concatenate([q, [0]]).astype(np.uint8))
        tt = np.ascontiguousarray(np.concatenate([t, [0]]).astype(np.uint8))
        na = ref.ref_global(int(q.size), C.c_void_p(qq.ctypes.data), int(t.size), C.c_void_p(tt.ctypes.data), 2, -5, 3, 1, 3, 1, w, C.byref(sa), C.c_void_p(ca.ctypes.data))
        nb = ora.ora_global_c(int(q.size), C.c_void_p(qq.ctypes.data), int(t.size), C.c_void_p(tt.ctypes.data), 2, -5, 3, 1, 3, 1, w, C.byref(sb), C.c_void_p(cb.ctypes.data))
        assert sa.value == sb.value and na == nb and (ca[:na] == cb[:nb]).all()


def test_unstable_sort_tie_order(libs):
    ref, ora = libs
    rng = np.random.default_rng(7)
    ref.ref_sort_u64_lo32_desc.argtypes = ora.ora_sort_u64_lo32_desc.argtypes = [C.c_void_p, C.c_size_t]
    ref.ref_sort_u32_asc.argtypes = ora.ora_sort_u32_asc.argtypes = [C.c_void_p, C.c_size_t]
    for n in list(range(0, 40)) + [100, 501, 1000, 5000]:
        for nkeys in (1, 3, 50, 1 << 20):
            v = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, nkeys, size=n).astype(np.uint64)
            a, b = v.copy(), v.copy()
            ref.ref_sort_u64_lo32_desc(a.ctypes.data, n)
            ora.ora_sort_u64_lo32_desc(b.ctypes.data, n)
            assert (a == b).all()
            w = rng.integers(0, nkeys, size=n).astype(np.uint32)
            a, b = w.copy(), w.copy()
            ref.ref_sort_u32_asc(a.ctypes.data, n)
            ora.ora_sort_u32_asc(b.ctypes.data, n)
            assert (a == b).all() and (np.diff(a.astype(np.int64)) >= 0).all()


def test_median(libs):
    ref, ora = libs
    rng = np.random.default_rng(8)
    for n in list(range(1, 30)) + [100, 1001]:
        v = rng.integers(-50, 50, size=n).astype(np.int32)
        a, b = v.copy(), v.copy()
        assert ref.ref_median(C.c_void_p(a.ctypes.data), n) == ora.ora_median_c(C.c_void_p(b.ctypes.data), n) == int(np.sort(v)[n // 2])
