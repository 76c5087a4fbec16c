"""Seeded synthetic PacBio-shape read generator.

genome  : iid uniform ACGT of length G
reads   : length ~ lognormal (mean `mean_len`, sigma 0.4), min `min_len`; uniform start;
          strand 50/50; per-base error `err` split ins:del:sub = 50:30:20; no N.

Everything is driven by numpy's PCG64 so the same (seed, parameters) give the same
bytes on every machine; fixtures under tests/golden/ record the md5 of the FASTA they
were generated from so generator drift is detected, not silently absorbed.
"""
from __future__ import annotations

import hashlib
import io
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def random_genome(G: int, rng: np.random.Generator) -> np.ndarray:
    """2-bit codes (0..3) of an iid uniform genome."""
    return rng.integers(0, 4, size=G, dtype=np.uint8)


def _mutate(seq: np.ndarray, err: float, rng: np.random.Generator) -> np.ndarray:
    """Apply ins:del:sub = 50:30:20 errors at total per-base rate `err`."""
    n = seq.size
    u = rng.random(n)
    p_del = err * 0.3
    p_sub = err * 0.2
    p_ins = err * 0.5
    keep = u >= p_del
    sub = (u >= p_del) & (u < p_del + p_sub)
    out = seq.copy()
    nsub = int(sub.sum())
    if nsub:
        out[sub] = (out[sub] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
    ins = rng.random(n) < p_ins
    # emitted length per source base: (1 if kept) + (1 if insertion before it)
    cnt = keep.astype(np.int64) + ins.astype(np.int64)
    total = int(cnt.sum())
    res = np.empty(total, dtype=np.uint8)
    pos = np.cumsum(cnt) - cnt  # start offset of each source base's emission
    ins_idx = pos[ins]
    res[ins_idx] = rng.integers(0, 4, size=ins_idx.size, dtype=np.uint8)
    keep_idx = (pos + ins.astype(np.int64))[keep]
    res[keep_idx] = out[keep]
    return res


def synth_reads(genome_len: int, coverage: float, seed: int = 1, mean_len: float = 10000.0,
                sigma: float = 0.4, min_len: int = 1000, err: float = 0.15,
                max_len: int | None = None, repeats: bool = False):
    """Returns (names, seqs) where seqs is a list of uint8 arrays of 2-bit codes.
    repeats: plant tandem arrays (300-bp unit x 40) and dispersed copies of a 6-kb element in the genome, so that pairs of reads
    have thousands of off-diagonal z-mer matches (the repeat-rich shape the iid genome never produces)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = random_genome(genome_len, rng)
    if repeats:
        r2 = np.random.Generator(np.random.PCG64(seed + 7919))
        unit = r2.integers(0, 4, size=300, dtype=np.uint8)
        elem = r2.integers(0, 4, size=6000, dtype=np.uint8)
        for k in range(max(2, genome_len // 60000)):
            st = int(r2.integers(0, max(1, genome_len - 12000)))
            arr = np.tile(unit, 40)[: min(12000, genome_len - st)]
            genome[st:st + arr.size] = arr
            st = int(r2.integers(0, max(1, genome_len - 6000)))
            genome[st:st + min(6000, genome_len - st)] = elem[: min(6000, genome_len - st)]
    mu = np.log(mean_len) - 0.5 * sigma * sigma
    target = int(genome_len * coverage)
    names, seqs = [], []
    tot = 0
    i = 0
    while tot < target:
        L = int(rng.lognormal(mu, sigma))
        if L < min_len:
            L = min_len
        if max_len is not None and L > max_len:
            L = max_len
        if L > genome_len:
            L = genome_len
        st = int(rng.integers(0, genome_len - L + 1))
        s = genome[st:st + L]
        if rng.random() < 0.5:
            s = _COMP[s[::-1]]
        r = _mutate(s, err, rng)
        names.append("pb%012d" % i)
        seqs.append(r)
        tot += r.size
        i += 1
    return names, seqs


def to_fasta_bytes(names, seqs, width: int = 0) -> bytes:
    buf = io.BytesIO()
    for n, s in zip(names, seqs):
        buf.write(b">" + n.encode() + b"\n")
        a = _ACGT[s]
        if width:
            for k in range(0, a.size, width):
                buf.write(a[k:k + width].tobytes() + b"\n")
        else:
            buf.write(a.tobytes() + b"\n")
    return buf.getvalue()


def write_fasta(path: str, names, seqs, width: int = 0) -> str:
    data = to_fasta_bytes(names, seqs, width)
    with open(path, "wb") as fh:
        fh.write(data)
    return hashlib.md5(data).hexdigest()


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("-G", type=int, default=300000, help="genome length")
    ap.add_argument("-c", type=float, default=20.0, help="coverage")
    ap.add_argument("-s", type=int, default=1, help="seed")
    ap.add_argument("-l", type=float, default=10000.0, help="mean read length")
    ap.add_argument("-e", type=float, default=0.15, help="error rate")
    ap.add_argument("-m", type=int, default=1000, help="min read length")
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    n, s = synth_reads(a.G, a.c, a.s, a.l, 0.4, a.m, a.e)
    print(write_fasta(a.o, n, s), len(n), sum(x.size for x in s))
