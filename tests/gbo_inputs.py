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


def pile_inputs(seed: int = 1023, n: int = 1100, L: int = 400):
    """A read side with MORE than SG_MAX_EDGE = 1 023 overlaps (wtlay.h:35): read r0000 and `n` error-free reads that all dovetail its 3' end, and a
    hand-written overlap file (the 16 columns of a wtzmo record) that lists r0000 against each of them and nothing else - no overlapper run needed.
    The reference's loader keeps the first 1 023 of them in file order (wtlay.h:459-463); with `-N 1` everything behind that is defined behaviour
    (the surplus is only written past a slice from iteration 2 on: DESIGN section 9), and the anchoring pass then aligns every pair among the kept
    1 023 neighbours - 522 753 pairs - so the output says exactly which edges survived the limit.  Returns (FASTA bytes, overlap-file bytes)."""
    rng = np.random.default_rng(seed)
    g = synth.random_genome(1000, rng)
    names, seqs, lines = ["r0000"], [g[0:L].copy()], []
    for k in range(1, n + 1):
        s = 100 + (k * 7) % 100
        names.append("r%04d" % k)
        seqs.append(g[s:s + L].copy())
        ol = L - s
        lines.append("\t".join(map(str, ["r0000", "+", L, s, L, "r%04d" % k, "+", L, 0, ol, 2 * ol, "1.000", ol, 0, 0, 0])))
    return synth.to_fasta_bytes(names, seqs), ("\n".join(lines) + "\n").encode()


def write_pile(dirname: str, expect=None):
    fa, ov = pile_inputs()
    got = {"md5_reads": hashlib.md5(fa).hexdigest(), "md5_ovl": hashlib.md5(ov).hexdigest()}
    if expect is not None:
        assert got["md5_reads"] == expect["md5_reads"] and got["md5_ovl"] == expect["md5_ovl"], "generator drift: pile inputs %r != %r" % (got, expect)
    pf, po = os.path.join(dirname, "pile.fa"), os.path.join(dirname, "pile.ovl16")
    open(pf, "wb").write(fa)
    open(po, "wb").write(ov)
    return pf, po, got
