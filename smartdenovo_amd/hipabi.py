"""ctypes binding of the C ABI in include/wtzmo_hip.h (libwtzmo_hip.so).

Python is test / benchmark orchestration only; the product is the C library + the C host driver.
Loading fails loudly when the HIP library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwtzmo_hip.so")

SYMBOLS = [
    "wtz_last_error", "wtz_device_count", "wtz_device_memory", "wtz_ctx_create", "wtz_ctx_destroy", "wtz_ctx_clone", "wtz_upload_reads",
    "wtz_index_build", "wtz_zindex_build", "wtz_candidates", "wtz_candidates_begin", "wtz_candidates_end", "wtz_batch_begin", "wtz_pairs_seed",
    "wtz_pairs_windows", "wtz_pairs_align", "wtz_fetch_cigars", "wtz_fetch_cigar_text", "wtz_fetch_cigar_text_begin", "wtz_fetch_cigar_text_end", "wtz_cigar_text_device", "wtz_host_alloc", "wtz_host_free", "wtz_get_counters", "wtz_reset_counters",
    "wtz_test_dp", "wtz_extend_batch", "wtz_pool_info", "wtz_pool_failure_kind",
    "wtz_index_count", "wtz_index_counts_fetch", "wtz_index_finish", "wtz_candidate_groups_begin", "wtz_candidate_groups_end", "wtz_candidate_groups_fetch", "wtz_cand_tail_host", "wtz_zindex_build_subset", "wtz_zindex_build_queries", "wtz_upload_reads_ascii", "wtz_fetch_read_bits", "wtz_append_revcomp_views",
]


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("ksize", "zsize", "hk", "hz", "ksave", "kovl", "ncand", "nbest",
                                          "kwin", "kstep", "ztot", "zovl", "max_kmer_freq", "max_zmer_freq", "max_kmer_var")] + \
               [("win_rep_norm", C.c_float), ("win_rep_cutoff", C.c_float)] + \
               [(n, C.c_int32) for n in ("w", "ew", "W", "M", "X", "O", "E", "T", "min_score")] + [("min_id", C.c_float)] + \
               [(n, C.c_int32) for n in ("dot_matrix", "xvar", "yvar", "min_block_len", "max_overhang")] + \
               [("deviation_penalty", C.c_float), ("gap_penalty", C.c_float), ("refine", C.c_int32), ("aux_strand", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        """wtzmo.c:1543-1588"""
        p = cls(ksize=16, zsize=10, hk=1, hz=1, ksave=4, kovl=300, ncand=500, nbest=100, kwin=800, kstep=400, ztot=300,
                zovl=200, max_kmer_freq=0, max_zmer_freq=64, max_kmer_var=2, win_rep_norm=20.0, win_rep_cutoff=100.0,
                w=50, ew=800, W=3200, M=2, X=-5, O=-3, E=-1, T=-50, min_score=200, min_id=0.5, dot_matrix=0, xvar=128,
                yvar=64, min_block_len=160, max_overhang=256, deviation_penalty=1.0, gap_penalty=0.05, refine=0, aux_strand=0)
        for k, v in kw.items():
            setattr(p, k, v)
        p.kstep = p.kwin // 2
        p.max_overhang = 2 * p.xvar
        return p


class IndexStats(C.Structure):
    _fields_ = [("n_occ", C.c_uint64), ("n_distinct", C.c_uint64), ("ktot", C.c_uint64), ("n_kept", C.c_uint64),
                ("max_kmer_freq", C.c_uint32), ("avg_rdlen", C.c_uint32)]


PAIR_SUMMARY = np.dtype([("n_hits", "<u4"), ("gate", "<u4"), ("ovl", "<u4", 2), ("nwin", "<u4", 2),
                         ("dm_score", "<i4"), ("dm_qb", "<i4"), ("dm_qe", "<i4"), ("dm_tb", "<i4"), ("dm_te", "<i4"), ("dm_dir", "<i4")])
WINBOX = np.dtype([("beg", "<i4", 2), ("end", "<i4", 2)])
ALN_RESULT = np.dtype([("score", "<i4"), ("tb", "<i4"), ("te", "<i4"), ("qb", "<i4"), ("qe", "<i4"), ("aln", "<i4"), ("mat", "<i4"),
                       ("mis", "<i4"), ("ins", "<i4"), ("del", "<i4"), ("n_regs", "<u4"), ("cigar_len", "<u4"), ("cigar_off", "<u8"),
                       ("text_len", "<u4"), ("pad", "<u4"), ("text_off", "<u8")])


DP_PROBLEM = np.dtype([("q_read", "<u4"), ("t_read", "<u4"), ("q_rev", "<u4"), ("t_rev", "<u4"), ("q_from", "<i4"), ("t_from", "<i4"),
                       ("q_strand", "<i4"), ("t_strand", "<i4"), ("q_len", "<i4"), ("t_len", "<i4"), ("init_score", "<i4"), ("W", "<i4")])
DP_RESULT = np.dtype([("score", "<i4"), ("tb", "<i4"), ("te", "<i4"), ("qb", "<i4"), ("qe", "<i4"), ("aln", "<i4"), ("mat", "<i4"),
                      ("mis", "<i4"), ("ins", "<i4"), ("del", "<i4"), ("cigar_len", "<u4"), ("form_used", "<u4"), ("cigar_off", "<u8"), ("cells", "<u8")])
DP_SHIFT, DP_FIXED, DP_GLOBAL = 0, 1, 2


class Counters(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_index", "ms_zindex", "ms_candidates", "ms_pairs", "ms_winalign", "ms_stitch")] + \
               [(n, C.c_uint64) for n in ("n_candidates_q", "n_pairs", "n_winalign", "n_stitch", "cells_shift", "cells_fixed",
                                          "cells_global", "bytes_seed_algo", "pool_peak")] + \
               [("ms_ext", C.c_double), ("n_extjobs", C.c_uint64), ("ms_gap", C.c_double), ("bytes_zmer_algo", C.c_uint64),
                ("ms_ingest", C.c_double), ("bytes_ingest_algo", C.c_uint64)]


def load(path: str | None = None) -> C.CDLL:
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("libwtzmo_hip.so not found at %s: build it with `python __graft_entry__.py build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path." % path)
    lib = C.CDLL(path)
    lib.wtz_last_error.restype = C.c_char_p
    lib.wtz_ctx_create.argtypes = [C.c_int, C.POINTER(Params), C.c_uint64, C.POINTER(C.c_void_p)]
    lib.wtz_ctx_destroy.argtypes = [C.c_void_p]
    lib.wtz_ctx_destroy.restype = None
    lib.wtz_upload_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.wtz_index_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(IndexStats)]
    lib.wtz_zindex_build.argtypes = [C.c_void_p]
    lib.wtz_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.wtz_batch_begin.argtypes = [C.c_void_p]
    lib.wtz_pairs_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.wtz_pairs_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.wtz_pairs_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.wtz_fetch_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.wtz_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    lib.wtz_reset_counters.argtypes = [C.c_void_p]
    lib.wtz_test_dp.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
    return lib


def check_symbols(path: str | None = None) -> None:
    """Every symbol declared in include/wtzmo_hip.h must be exported (no compute call: works without a GPU)."""
    lib = C.CDLL(path or LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError("libwtzmo_hip.so does not export: " + ", ".join(missing))


def pack_reads(seqs):
    """2-bit pack (dna.h:78 layout) a list of uint8 code arrays given in READ-ID order."""
    lens = np.array([s.size for s in seqs], dtype=np.uint32)
    offs = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    allb = np.concatenate(seqs).astype(np.uint64) if len(seqs) else np.zeros(0, dtype=np.uint64)
    n = allb.size
    nw = (n + 31) // 32
    pad = np.zeros(nw * 32, dtype=np.uint64)
    pad[:n] = allb
    pad = pad.reshape(nw, 32)
    shifts = (np.uint64(62) - np.arange(32, dtype=np.uint64) * np.uint64(2))
    words = np.bitwise_or.reduce(pad << shifts, axis=1) if nw else np.zeros(0, dtype=np.uint64)
    return np.ascontiguousarray(words, dtype=np.uint64), offs, lens


class Context:
    """Thin RAII wrapper used by tests and bench.py."""

    def __init__(self, params: Params, device: int = 0, pool_bytes: int = 0, lib_path: str | None = None):
        self.lib = load(lib_path)
        self.params = params
        self.h = C.c_void_p()
        self._chk(self.lib.wtz_ctx_create(device, C.byref(params), pool_bytes, C.byref(self.h)))

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libwtzmo_hip error %d: %s" % (rc, self.lib.wtz_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.wtz_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def upload(self, words, offs, lens):
        self.n_reads = int(lens.size)
        self._keep = (words, offs, lens)
        self._chk(self.lib.wtz_upload_reads(self.h, words.ctypes.data, words.size, offs.ctypes.data, lens.ctypes.data, lens.size))

    def upload_ascii(self, text: bytes, offs, lens, rand_calls_before=0):
        """f4: the bases as text, packed on the device (wtz_upload_reads_ascii); returns the number of non-ACGT bytes"""
        self.n_reads = int(lens.size)
        self._keep = (text, offs, lens)
        nr = C.c_uint64(0)
        self.lib.wtz_upload_reads_ascii.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
        self._chk(self.lib.wtz_upload_reads_ascii(self.h, text, len(text), offs.ctypes.data, lens.ctypes.data, lens.size, rand_calls_before, C.byref(nr)))
        return int(nr.value)

    def append_revcomp_views(self):
        """reads n .. 2n-1 := reverse complements of reads 0 .. n-1 (wtz_append_revcomp_views)"""
        self.lib.wtz_append_revcomp_views.argtypes = [C.c_void_p]
        self._chk(self.lib.wtz_append_revcomp_views(self.h))
        self.n_reads *= 2

    def fetch_read_bits(self, n_bases):
        nw = (n_bases + 31) // 32
        out = np.zeros(nw, dtype=np.uint64)
        self.lib.wtz_fetch_read_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._chk(self.lib.wtz_fetch_read_bits(self.h, out.ctypes.data, nw))
        return out

    def index_build(self, beg=0, end=None, K=0):
        k = C.c_uint32(K)
        st = IndexStats()
        self._chk(self.lib.wtz_index_build(self.h, beg, self.n_reads if end is None else end, C.byref(k), C.byref(st)))
        return st

    def zindex_build(self):
        self._chk(self.lib.wtz_zindex_build(self.h))

    def candidates(self, qids):
        qids = np.ascontiguousarray(qids, dtype=np.uint32)
        stride = self.params.ncand + 1
        rows = np.zeros((qids.size, stride), dtype=np.uint64)
        n = np.zeros(qids.size, dtype=np.uint32)
        self._chk(self.lib.wtz_candidates(self.h, qids.ctypes.data, qids.size, rows.ctypes.data, n.ctypes.data))
        return rows, n

    def pairs_seed(self, q, c):
        q = np.ascontiguousarray(q, dtype=np.uint32)
        c = np.ascontiguousarray(c, dtype=np.uint32)
        out = np.zeros(q.size, dtype=PAIR_SUMMARY)
        self._chk(self.lib.wtz_pairs_seed(self.h, q.ctypes.data, c.ctypes.data, q.size, out.ctypes.data))
        return out

    def pairs_windows(self, summary):
        n = int(summary["nwin"].sum())
        out = np.zeros(n, dtype=WINBOX)
        self._chk(self.lib.wtz_pairs_windows(self.h, out.ctypes.data, n))
        return out

    def pairs_align(self, pair_idx, dirs):
        pair_idx = np.ascontiguousarray(pair_idx, dtype=np.uint32)
        dirs = np.ascontiguousarray(dirs, dtype=np.uint8)
        out = np.zeros(pair_idx.size, dtype=ALN_RESULT)
        self._chk(self.lib.wtz_pairs_align(self.h, pair_idx.ctypes.data, dirs.ctypes.data, pair_idx.size, out.ctypes.data))
        tot = int(out["cigar_len"].sum())
        cig = np.zeros(tot, dtype=np.uint32)
        self._chk(self.lib.wtz_fetch_cigars(self.h, cig.ctypes.data, tot))
        return out, cig

    def test_dp(self, kind, form, problems, cigar_cap=1 << 22):
        """TEST-ONLY: run DP problems through one device form; returns (results, list of CIGAR arrays)."""
        problems = np.ascontiguousarray(problems, dtype=DP_PROBLEM)
        out = np.zeros(problems.size, dtype=DP_RESULT)
        cig = np.zeros(cigar_cap, dtype=np.uint32)
        self._chk(self.lib.wtz_test_dp(self.h, kind, form, problems.ctypes.data, problems.size, out.ctypes.data, cig.ctypes.data, cig.size))
        return out, [cig[int(o):int(o) + int(n)].copy() for o, n in zip(out["cigar_off"], out["cigar_len"])]

    def counters(self) -> Counters:
        c = Counters()
        self._chk(self.lib.wtz_get_counters(self.h, C.byref(c)))
        return c

    def reset_counters(self):
        self._chk(self.lib.wtz_reset_counters(self.h))
