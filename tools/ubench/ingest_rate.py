"""f4 measurement: FASTA text -> 2-bit BaseBank.  Device: wtz_upload_reads_ascii of 2^30 random ACGT bytes (HIP-event kernel time from the library's
counters).  CPU beside it: the reference's own loader seq2basebank (dna.h:397-410) through oracle/_ref/libref_shim.so on 2^28 bytes, one thread
(the reference loads reads on its main thread)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smartdenovo_amd import hipabi  # noqa: E402

n = 1 << 30
rng = np.random.default_rng(1)
text = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)].tobytes()
lens = np.array([1000], dtype=np.uint32)
offs = np.zeros(1, dtype=np.uint64)
ctx = hipabi.Context(hipabi.Params.defaults(), pool_bytes=1 << 28)
for rep in range(3):
    ctx.reset_counters()
    t0 = time.time()
    ctx.upload_ascii(text, offs, lens)
    t1 = time.time()
    c = ctx.counters()
    print("rep %d: %d bases, kernels %.3f ms = %.0f GB/s algorithmic; call %.0f ms" % (rep, n, c.ms_ingest, c.bytes_ingest_algo / c.ms_ingest / 1e6, 1e3 * (t1 - t0)))
ctx.close()
shim = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
if os.path.exists(shim):
    lib = C.CDLL(shim)
    lib.ref_seq2basebank.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
    m = 1 << 28
    out = np.zeros(m // 32 + 2, dtype=np.uint64)
    t0 = time.time()
    lib.ref_seq2basebank(text[:m], m, 0, out.ctypes.data)
    dt = time.time() - t0
    print("reference seq2basebank on this host, 1 thread: %d bases in %.3f s = %.2f GB/s algorithmic (1.25 B per base)" % (m, dt, 1.25 * m / dt / 1e9))
