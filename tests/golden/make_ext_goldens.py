#!/usr/bin/env python3
"""Golden fixtures of the `wtext` drop-in, made with the REAL reference binaries (oracle/_ref/wtext_ref, wtzmo_ref, wtobt_ref:
`make -C oracle ref`).  Build container only.  Always `-t 1`.

Inputs.  The 17-column overlap files are NOT stored (the CIGAR column is most of their bytes): the tests regenerate them with wtzmo (oracle on
the CPU, bin/wtzmo on the GPU) and check the md5 recorded here before they use them.  Stored: the retained-region files - ext_tiny.obt is the
reference wtobt's output on the tiny case, ext_hard.clp / ext_brutal.clp / ext_prev.clp are seeded random regions (a comment line, an unknown
read and out-of-range regions included: wtext.c:441-461 skips them).
Outputs (committed): ext_manifest.json - per case argv, md5 of the full 17-column output, record count; ext_<case>.ovl16.gz - first 16 columns.
Fixtures are data (inputs + expected outputs); no reference source text is stored."""
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")

# overlap inputs: name -> wtzmo argv ('@x' = tests/golden/x)
OVLS = {
    "zmo": ["-k", "16", "-s", "200", "-m", "0.6"],
    "dmo": ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"],
    "zmo_prev": ["-k", "16", "-s", "200", "-m", "0.6", "-b", "@ext_prev.clp"],
}
# name, overlap inputs, extra argv
CASES = [
    ("obt", ["zmo"], ["-b", "@ext_tiny.obt"]),
    ("none", ["zmo"], []),
    ("hard", ["zmo"], ["-b", "@ext_hard.clp"]),
    ("brutal", ["zmo"], ["-b", "@ext_brutal.clp"]),
    ("prev", ["zmo_prev"], ["-B", "@ext_prev.clp"]),
    ("prev_hard", ["zmo_prev"], ["-B", "@ext_prev.clp", "-b", "@ext_hard.clp"]),
    ("scores", ["zmo"], ["-b", "@ext_hard.clp", "-W", "100", "-M", "3", "-X", "-4", "-O", "-2", "-E", "-2", "-T", "-20"]),
    ("T0_W30", ["zmo"], ["-b", "@ext_tiny.obt", "-T", "0", "-W", "30"]),
    ("P2p0", ["zmo"], ["-b", "@ext_hard.clp", "-P", "2", "-p", "0"]),
    ("P2p1", ["zmo"], ["-b", "@ext_hard.clp", "-P", "2", "-p", "1"]),
    ("dmo_hard", ["dmo"], ["-b", "@ext_hard.clp"]),
    ("two_files", ["zmo", "dmo"], ["-b", "@ext_tiny.obt", "-b", "@ext_hard.clp"]),
]


def md5(b):
    return hashlib.md5(b).hexdigest()


def read_lengths(fa):
    names, lens = [], []
    for l in open(fa):
        if l[0] == ">":
            names.append(l[1:].split()[0]); lens.append(0)
        else:
            lens[-1] += len(l.strip())
    return names, lens


def write_regions(fa):
    names, lens = read_lengths(fa)
    rng = np.random.default_rng(7)
    with open(os.path.join(HERE, "ext_hard.clp"), "w") as f:
        f.write("# a comment line\n")
        for n, L in zip(names, lens):
            r = rng.random()
            if r < 0.15:
                continue
            a = int(rng.integers(0, min(2500, L // 3))); b = int(rng.integers(0, min(2500, L // 3)))
            if r < 0.2:
                a = 0
            if r > 0.95:
                f.write("%s\t%d\t%d\n" % (n, a, L + 5)); continue          # beyond the read: ignored
            f.write("%s\t%d\t%d\n" % (n, a, L - a - b))
        f.write("nosuchread\t0\t10\n")
    with open(os.path.join(HERE, "ext_prev.clp"), "w") as f:
        for n, L in zip(names, lens):
            if rng.random() < 0.5:
                continue
            a = int(rng.integers(0, 400)); b = int(rng.integers(0, 400))
            f.write("%s\t%d\t%d\n" % (n, a, L - a - b))
    rng = np.random.default_rng(11)
    with open(os.path.join(HERE, "ext_brutal.clp"), "w") as f:
        for n, L in zip(names, lens):
            a = int(rng.integers(0, L * 6 // 10)); b = int(rng.integers(0, L - a))
            if rng.random() < 0.1:
                f.write("%s\t%d\t%d\n" % (n, a, 0)); continue
            f.write("%s\t%d\t%d\n" % (n, a, L - a - b))


def main():
    man = {"reference": "wtext -t 1 (oracle/_ref/wtext_ref)", "reads": "tiny.fa.gz", "ovls": {}, "cases": {}}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "tiny.fa")
        open(fa, "wb").write(gzip.open(os.path.join(HERE, "tiny.fa.gz")).read())
        write_regions(fa)
        for name, argv in OVLS.items():
            o = os.path.join(td, name + ".ovl")
            av = [a if not a.startswith("@") else os.path.join(HERE, a[1:]) for a in argv]
            subprocess.run([os.path.join(REF, "wtzmo_ref"), "-t", "1", "-i", fa, "-fo", o] + av, check=True, stderr=subprocess.DEVNULL)
            b = open(o, "rb").read()
            man["ovls"][name] = {"argv": argv, "md5_full": md5(b), "records": b.count(b"\n")}
        subprocess.run([os.path.join(REF, "wtobt_ref"), "-i", fa, "-j", os.path.join(td, "zmo.ovl"), "-fo", os.path.join(HERE, "ext_tiny.obt"), "-m", "0.6", "-c", "2"],
                       check=True, stderr=subprocess.DEVNULL)
        for name, ovls, extra in CASES:
            out = os.path.join(td, "x_" + name + ".ovl")
            av = [a if not a.startswith("@") else os.path.join(HERE, a[1:]) for a in extra]
            js = []
            for o in ovls:
                js += ["-j", os.path.join(td, o + ".ovl")]
            subprocess.run([os.path.join(REF, "wtext_ref"), "-t", "1", "-i", fa] + js + ["-fo", out] + av, check=True, stderr=subprocess.DEVNULL)
            full = open(out, "rb").read()
            lines = [l for l in full.split(b"\n") if l]
            cut = b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in lines) + (b"\n" if lines else b"")
            with gzip.GzipFile(os.path.join(HERE, "ext_%s.ovl16.gz" % name), "wb", mtime=0) as f:
                f.write(cut)
            man["cases"][name] = {"ovls": ovls, "argv": extra, "md5_full": md5(full), "records": len(lines)}
            print(name, len(lines), md5(full))
    json.dump(man, open(os.path.join(HERE, "ext_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
