"""Seeded inputs of the wtgbo parity tests that are generated instead of stored (the FASTA of the `grid` case).

grid reads: starts on a coarse grid of an iid genome, lengths in steps of 500, half of them error-free, 30 % reverse
strand — many reads share a start, so that one read is a prefix of another and the reference's "contained" break
(wtgbo.c:51-54, 190-191: the hit starts at 0 and ends at the candidate's length) and its repeated output line occur."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartdenovo_amd import synth  # noqa: E402


def grid_fasta(seed: int, G: int = 200000, cov: float = 15.0, err: float = 0.10, grid: int = 1000) -> bytes:
    rng = np.random.default_rng(seed)
    g = synth.random_genome(G, rng)
    names, seqs = [], []
    tot = 0
    while tot < G * cov:
        L = int(rng.integers(3, 10)) * 1000 + int(rng.integers(0, 3)) * 500
        s = int(rng.integers(0, (G - L) // grid + 1)) * grid
        seq = g[s:s + L].copy()
        if err > 0 and rng.random() < 0.5:
            seq = synth._mutate(seq, err, rng)
        if rng.random() < 0.3:
            seq = (3 - seq)[::-1]
        names.append("g%06d" % len(names))
        seqs.append(seq)
        tot += L
    return synth.to_fasta_bytes(names, seqs)


def write_grid(path: str, seed: int, expect_md5: str = None) -> str:
    b = grid_fasta(seed)
    m = hashlib.md5(b).hexdigest()
    if expect_md5 is not None:
        assert m == expect_md5, "generator drift: grid(%d) md5 %s != %s" % (seed, m, expect_md5)
    open(path, "wb").write(b)
    return m
