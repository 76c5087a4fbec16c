"""f4: FASTA bases -> the reference's 2-bit BaseBank on the device (wtz_upload_reads_ascii).
The checker is the reference's own seq2basebank (dna.h:397-410) called through the shim from the lrand48 state of a fresh process: every word of the
bank must be equal, including the words that hold non-ACGT bytes (the reference draws `lrand48() & 3` for each, in file order)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SHIM), reason="reference shim not built (needs /root/reference once)")


def make_text(seed, n, n_other):
    rng = np.random.default_rng(seed)
    t = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=n)].copy()
    if n_other:
        pos = rng.choice(n, size=n_other, replace=False)
        t[pos] = np.frombuffer(b"NnRYKMSWBDHVU-*.X", dtype=np.uint8)[rng.integers(0, 17, size=n_other)]
        r0 = int(rng.integers(0, max(1, n - 3000)))
        t[r0:r0 + min(2500, n - r0)] = ord("N")          # a long run: many draws inside consecutive words
    return t.tobytes()


def ref_bits(text, skip=0):
    lib = C.CDLL(SHIM)
    lib.ref_seq2basebank.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
    out = np.zeros((len(text) + 31) // 32 + 1, dtype=np.uint64)
    lib.ref_seq2basebank(text, len(text), skip, out.ctypes.data)
    return out[:(len(text) + 31) // 32]


def device_bits(lib_path, text, skip=0):
    from smartdenovo_amd import hipabi
    ctx = hipabi.Context(hipabi.Params.defaults(), pool_bytes=1 << 28, lib_path=lib_path)
    try:
        lens = np.array([len(text)], dtype=np.uint32); offs = np.zeros(1, dtype=np.uint64)
        n_other = ctx.upload_ascii(text, offs, lens, rand_calls_before=skip)
        return ctx.fetch_read_bits(len(text)), n_other, ctx.counters()
    finally:
        ctx.close()


@pytest.fixture(scope="module")
def emul_lib():
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return os.path.join(ROOT, "tests", "emul", "libwtz_emul.so")


@pytest.mark.parametrize("n,n_other,skip", [(1, 0, 0), (31, 3, 0), (32, 0, 0), (33, 33, 0), (100003, 0, 0), (100003, 700, 0), (250000, 5000, 12345), (4097, 4097, 7)])
def test_emulated_ingest_equals_reference_loader(n, n_other, skip, emul_lib):
    text = make_text(n * 7 + n_other, n, min(n_other, n))
    if n_other >= n:
        text = b"N" * n
    want = ref_bits(text, skip)
    got, cnt, _ = device_bits(emul_lib, text, skip)
    assert cnt == sum(1 for ch in text if ch not in b"ACGTacgt")
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_other,skip", [(33, 5, 0), (1000003, 0, 0), (64 * 1024 * 1024 + 17, 200000, 3), (300 * 1024 * 1024 + 5, 1500000, 0)])
def test_gpu_ingest_equals_reference_loader(n, n_other, skip):
    """the last case spans two device chunks (256 M bases each) and overflows the first list of non-base positions (1 M entries)"""
    text = make_text(n + n_other, n, n_other)
    want = ref_bits(text, skip)
    got, cnt, c = device_bits(None, text, skip)
    assert np.array_equal(got, want)
    if n > 1 << 20:
        print("ingest %d bases: kernel %.3f ms, %.1f GB/s algorithmic" % (n, c.ms_ingest, c.bytes_ingest_algo / c.ms_ingest / 1e6))


def test_first_draws_are_those_of_a_fresh_glibc_process(emul_lib):
    """0, 2116118, 89401895, 379337186: `lrand48() & 3` of a never-seeded process = 0, 2, 3, 2 - independent of the shim"""
    got, cnt, _ = device_bits(emul_lib, b"NNNNACGT")
    assert cnt == 4 and int(got[0]) >> 48 == int("00" "10" "11" "10" "00" "01" "10" "11", 2)


# ---- the host driver: both ingest paths write the reference's file (edge.fq holds runs of N / n: the draws matter) ----
@pytest.mark.parametrize("mode", ["device", "host"])
def test_wtzmo_ingest_modes_on_emulated_device(mode, tmp_path):
    from conftest import manifest, run_wtzmo_like
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    case = manifest()["cases"]["zmo_edge_fq"]
    md5, cont, _ = run_wtzmo_like(os.path.join(ROOT, "tests", "emul", "wtzmo_emul"), case, tmp_path, extra=["--ingest", mode, "--batch", "16"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("name", ["zmo_edge_fq", "dmo_edge_fa"])
def test_gpu_wtzmo_ingest_modes(mode, name, gpu_exe, tmp_path):
    from conftest import manifest, run_wtzmo_like
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(gpu_exe, case, tmp_path, extra=["--ingest", mode])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


def test_revcomp_views_on_emulated_device(emul_lib):
    """wtz_append_revcomp_views: read n + i = reverse complement of read i (revbitseq_basebank), packed on the device; ragged lengths, one empty read"""
    from smartdenovo_amd import hipabi
    rng = np.random.default_rng(9)
    seqs = [rng.integers(0, 4, size=int(L), dtype=np.uint8) for L in (1, 31, 32, 33, 64, 1000, 4097, 0, 77)]
    ctx = hipabi.Context(hipabi.Params.defaults(), pool_bytes=1 << 28, lib_path=emul_lib)
    try:
        words, offs, lens = hipabi.pack_reads(seqs)
        ctx.upload(words, offs, lens)
        ctx.append_revcomp_views()
        nold = int(words.size)
        tot_new = sum((s.size + 31) // 32 for s in seqs)
        bits = ctx.fetch_read_bits((nold + tot_new) * 32)
        at = nold
        for s in seqs:
            nw = (s.size + 31) // 32
            want, _, _ = hipabi.pack_reads([(3 - s)[::-1].copy()]) if s.size else (np.zeros(0, dtype=np.uint64), None, None)
            assert np.array_equal(bits[at:at + nw], want[:nw])
            at += nw
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_wtgbo_ingest_modes(tmp_path):
    """bin/wtgbo packs the reads and makes their reverse-complement views on the device by default; --ingest host packs while reading: same bytes (tiny_b_L has clipped reads)"""
    import test_wtgbo as G
    import __graft_entry__ as ge
    for name in ("tiny_b_L", "grid4"):
        for mode in ("device", "host"):
            G.check_case(ge.EXE_GBO, name, tmp_path, ["--ingest", mode])
