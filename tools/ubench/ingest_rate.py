import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from smartdenovo_amd import hipabi
n = 1 << 30
rng = np.random.default_rng(1)
text = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)].tobytes()
lens = np.array([n & 0xFFFFFFFF or 1], dtype=np.uint32); lens[0] = 1000; offs = np.zeros(1, dtype=np.uint64)
ctx = hipabi.Context(hipabi.Params.defaults(), pool_bytes=1 << 28)
for rep in range(3):
    ctx.reset_counters()
    t0 = time.time(); ctx.upload_ascii(text, offs, lens); t1 = time.time()
    c = ctx.counters()
    print("rep %d: %d bases, kernels %.3f ms = %.0f GB/s algorithmic; call %.0f ms" % (rep, n, c.ms_ingest, c.bytes_ingest_algo / c.ms_ingest / 1e6, 1e3 * (t1 - t0)))
ctx.close()
