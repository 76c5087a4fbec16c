#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the REAL reference binary.

Run in the build container only (needs oracle/_ref/wtzmo_ref, i.e. `make -C oracle ref`,
which compiles /root/reference's own sources where they lie).  Always `-t 1`: the
reference is deterministic only with one worker (SURVEY.md finding 1).

Outputs (all committed):
  tiny.fa.gz / edge.fa.gz / edge.fq.gz     small inputs (seeded synthetic, see smartdenovo_amd/synth.py)
  <case>.ovl16.gz                          first 16 columns of the reference .ovl
  manifest.json                            per case: argv, md5 of the FULL .ovl (incl. CIGAR),
                                           md5 of .contained, record count, md5 of the input
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from smartdenovo_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")

ZMO = ["-k", "16", "-s", "200", "-m", "0.6"]
DMO = ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"]

CASES = [
    # name, input, extra argv
    ("zmo", "tiny.fa.gz", ZMO),
    ("dmo", "tiny.fa.gz", DMO),
    ("zmo_C", "tiny.fa.gz", ZMO + ["-C"]),
    # the reference reads past the read table unless n_rd % n_idx == 0 (wtzmo.c:1283 vs 253): -J @even drops
    # the shortest read(s) so that the count is even (resolved in make_inputs)
    ("zmo_G2", "tiny.fa.gz", ZMO + ["-G", "2", "-J", "@even"]),
    ("dmo_G2", "tiny.fa.gz", DMO + ["-G", "2", "-J", "@even"]),
    ("zmo_P2p0", "tiny.fa.gz", ZMO + ["-P", "2", "-p", "0"]),
    ("zmo_P2p1", "tiny.fa.gz", ZMO + ["-P", "2", "-p", "1"]),
    ("zmo_H0", "tiny.fa.gz", ZMO + ["-H", "0"]),
    ("zmo_H1", "tiny.fa.gz", ZMO + ["-H", "1"]),
    ("zmo_k17", "tiny.fa.gz", ["-k", "17", "-s", "200", "-m", "0.6"]),
    ("zmo_S1", "tiny.fa.gz", ZMO + ["-S", "1"]),
    ("zmo_K12", "tiny.fa.gz", ZMO + ["-K", "12"]),
    ("zmo_d600_r500_q30_l0", "tiny.fa.gz", ZMO + ["-d", "600", "-r", "500", "-q", "30", "-l", "0"]),
    ("zmo_N", "tiny.fa.gz", ZMO + ["-N"]),
    ("zmo_A5", "tiny.fa.gz", ZMO + ["-A", "5"]),
    ("zmo_B2", "tiny.fa.gz", ZMO + ["-B", "2"]),
    ("zmo_J4000", "tiny.fa.gz", ZMO + ["-J", "4000"]),
    ("zmo_w20_e200", "tiny.fa.gz", ZMO + ["-w", "20", "-e", "200", "-W", "800"]),
    ("zmo_scores", "tiny.fa.gz", ZMO + ["-M", "3", "-X", "-4", "-O", "-2", "-E", "-2", "-T", "-20"]),
    ("zmo_z12", "tiny.fa.gz", ZMO + ["-z", "12", "-Z", "32", "-y", "600", "-R", "150", "-r", "250"]),
    ("dmo_U", "tiny.fa.gz", ["-k", "16", "-z", "10", "-Z", "16", "-U", "128", "-U", "64", "-U", "160", "-U", "1.0", "-U", "0.05", "-m", "0.1", "-A", "1000"]),
    ("dmo_U2", "tiny.fa.gz", ["-k", "16", "-z", "10", "-Z", "16", "-U", "96", "-U", "48", "-U", "120", "-U", "0.5", "-U", "0.1", "-m", "0.1", "-A", "50"]),
    ("dmo_A5", "tiny.fa.gz", DMO[:-2] + ["-A", "5"]),
    ("zmo_edge_fa", "edge.fa.gz", ZMO),
    ("dmo_edge_fa", "edge.fa.gz", DMO),
    ("zmo_edge_fq", "edge.fq.gz", ZMO),
    ("zmo_L", "tiny.fa.gz", ZMO + ["-L", "@pairs.txt"]),
    ("zmo_F", "tiny.fa.gz", ZMO + ["-F", "@mask.txt"]),
    ("zmo_b", "tiny.fa.gz", ZMO + ["-b", "@clips.txt"]),
    ("zmo_n", "tiny.fa.gz", ZMO + ["-n"]),                                   # A11: kswx_refine_alignment
    ("zmo_n_w20", "tiny.fa.gz", ZMO + ["-n", "-w", "20", "-M", "3", "-X", "-4"]),
    # -N with the dot-matrix engine: print_hits_wtzmo walks the (empty) seed list, so the .ovl is EMPTY (wtzmo.c:1175-1210, 1319)
    ("dmo_N", "tiny.fa.gz", DMO + ["-N"]),
    # -I: query-only reads (wtzmo.c:1714-1729): the n_qr path, avg_rdlen over the query reads (361-368), query range 1304-1308
    ("zmo_I", "tiny.fa.gz", ZMO + ["-I", "@edge.fa.gz"]),
    ("dmo_I", "tiny.fa.gz", DMO + ["-I", "@edge.fa.gz"]),
    # -9: the tested-pairs file (wtzmo.c:1793-1804); the reference lists it in hash-table order, so the SET is pinned (md5 of the sorted lines)
    ("zmo_9", "tiny.fa.gz", ZMO + ["-9", "@out:pairs"]),
    ("dmo_9", "tiny.fa.gz", DMO + ["-9", "@out:pairs"]),
]


def md5(b: bytes) -> str:
    return hashlib.md5(b).hexdigest()


def make_inputs():
    names, seqs = synth.synth_reads(40000, 25, seed=7, mean_len=8000.0, min_len=1500)
    fa = synth.to_fasta_bytes(names, seqs)
    with gzip.GzipFile(os.path.join(HERE, "tiny.fa.gz"), "wb", mtime=0) as fh:
        fh.write(fa)
    # edge input: N runs, lower case, equal-length reads, an exact duplicate, wrapped lines, header comments
    import numpy as np
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(60):
        s = acgt[seqs[i]].tobytes()
        if i % 7 == 0:
            s = s[:100] + b"N" * 13 + s[113:2000] + b"nnnn" + s[2004:]
        if i % 5 == 0:
            s = s.lower()
        if i in (10, 11, 12, 13):
            s = s[:3000]                      # equal lengths -> unstable-sort tie order
        recs.append((names[i].encode() + (b" some comment" if i % 3 == 0 else b""), s))
    recs.append((b"dup_of_3", recs[3][1]))    # exact duplicate read
    recs.append((b"rc_like", recs[4][1][::-1]))
    buf = []
    for n, s in recs:
        buf.append(b">" + n + b"\n")
        for k in range(0, len(s), 70):
            buf.append(s[k:k + 70] + b"\n")
    with gzip.GzipFile(os.path.join(HERE, "edge.fa.gz"), "wb", mtime=0) as fh:
        fh.write(b"".join(buf))
    buf = []
    for n, s in recs:
        buf.append(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    with gzip.GzipFile(os.path.join(HERE, "edge.fq.gz"), "wb", mtime=0) as fh:
        fh.write(b"".join(buf))
    lens = sorted(x.size for x in seqs)
    EVEN_J[0] = lens[0] + 1 if len(lens) % 2 else 0
    assert sum(1 for x in lens if x >= EVEN_J[0]) % 2 == 0
    # side inputs
    # -L pairs: every third pair the reference itself reports on the zmo run, so the preload bites
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ovl")
        subprocess.run([REF, "-t", "1", "-i", os.path.join(HERE, "tiny.fa.gz"), "-fo", out] + ZMO, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        recs_ = [l.split("\t") for l in open(out) if l.strip()]
    with open(os.path.join(HERE, "pairs.txt"), "w") as fh:
        for r in recs_[::3]:
            fh.write("%s\t%s\n" % (r[5], r[0]))
        fh.write("# comment\nnot_a_read\t%s\n" % names[0])
    with open(os.path.join(HERE, "mask.txt"), "w") as fh:
        for i in range(0, len(names), 9):
            fh.write(names[i] + "\n")
    with open(os.path.join(HERE, "clips.txt"), "w") as fh:
        for i in range(0, len(names), 4):
            fh.write("%s\t%d\t%d\n" % (names[i], 100 + i, max(500, seqs[i].size - 300 - 2 * i)))


EVEN_J = [0]


def run_case(name, inp, extra):
    extra = [a if a != "@even" else str(EVEN_J[0]) for a in extra]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ovl")
        argv = [os.path.join(td, a[5:]) if a.startswith("@out:") else (a if not a.startswith("@") else os.path.join(HERE, a[1:])) for a in extra]
        cmd = [REF, "-t", "1", "-i", os.path.join(HERE, inp), "-fo", out] + argv
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        full = open(out, "rb").read()
        cont_path = out + ".contained"
        cont = open(cont_path, "rb").read() if os.path.exists(cont_path) else None
        pairs_path = os.path.join(td, "pairs")
        pairs_raw = open(pairs_path, "rb").read() if os.path.exists(pairs_path) else None
        pairs = sorted(pairs_raw.split(b"\n")) if pairs_raw is not None else None
    lines = full.split(b"\n")
    cut = b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in lines)
    with gzip.GzipFile(os.path.join(HERE, name + ".ovl16.gz"), "wb", mtime=0) as fh:
        fh.write(cut)
    return {
        "input": inp,
        "argv": extra,
        "records": sum(1 for l in lines if l and not l.startswith(b"#")),
        "lines": len(lines) - 1,
        "md5_full": md5(full),
        "md5_contained": md5(cont) if cont is not None else None,
        "contained": cont.decode().split() if cont is not None else None,
        "md5_pairs_sorted": md5(b"\n".join(pairs)) if pairs is not None else None,
        "md5_pairs": md5(pairs_raw) if pairs_raw is not None else None,       # the reference's own order (iteration order of its hash set)
        "pairs": len([p for p in pairs if p]) if pairs is not None else None,
    }


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    make_inputs()
    man = {"reference": "ruanjue/smartdenovo wtzmo, built by oracle/Makefile, run with -t 1", "inputs": {}, "cases": {}}
    for f in ("tiny.fa.gz", "edge.fa.gz", "edge.fq.gz"):
        man["inputs"][f] = md5(gzip.open(os.path.join(HERE, f)).read())
    for name, inp, extra in CASES:
        man["cases"][name] = run_case(name, inp, extra)
        print(name, man["cases"][name]["records"], man["cases"][name]["md5_full"])
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
