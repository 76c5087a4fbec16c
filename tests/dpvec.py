"""Loader of tests/golden/dp_vectors.npz (function-level vectors dumped from the reference's own DP routines by
tests/golden/make_dp_vectors.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Vectors:
    def __init__(self):
        z = np.load(os.path.join(HERE, "golden", "dp_vectors.npz"))
        self.kind, self.cls, self.init, self.W, self.w_param = z["kind"], z["cls"], z["init"], z["W"], z["w_param"]
        self.qlen, self.tlen, self.view = z["qlen"], z["tlen"], z["view"]
        self.read_off, self.read_codes = z["read_off"], z["read_codes"]
        self.aln, self.cig_off, self.cig = z["aln"], z["cig_off"], z["cig"]
        self.meta = json.loads(str(z["meta"]))
        self.n = self.kind.size

    def read(self, r):
        return self.read_codes[self.read_off[r]:self.read_off[r + 1]]

    def reads(self):
        return [self.read(r) for r in range(2 * self.n)]

    def logical(self, i, side):
        """the problem's sequence as the DP sees it (view resolved)"""
        rev, frm, strand = (int(x) for x in self.view[i][(0 if side == "q" else 3):(3 if side == "q" else 6)])
        n = int(self.qlen[i] if side == "q" else self.tlen[i])
        v = self.read(2 * i + (0 if side == "q" else 1))
        if rev:
            v = (3 - v)[::-1]
        if n == 0:
            return np.zeros(0, np.uint8)
        return np.ascontiguousarray(v[frm:frm + n] if strand > 0 else v[frm - n + 1:frm + 1][::-1])

    def expected_cigar(self, i):
        return self.cig[self.cig_off[i]:self.cig_off[i + 1]]
