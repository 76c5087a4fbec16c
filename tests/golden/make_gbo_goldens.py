#!/usr/bin/env python3
"""Golden fixtures of the `wtgbo` drop-in, made with the REAL reference binaries
(oracle/_ref/wtgbo_ref, oracle/_ref/wtzmo_ref: `make -C oracle ref`).  Build container only.  Always `-t 1`.

Outputs (committed): gbo_manifest.json — per case argv, md5 of the full 17-column output, md5 of the -9 pair file (the reference's own
hash-slot order), record count, number of repeated lines (the contained-break quirk); gbo_<case>.ovl16.gz — first 16 columns;
grid4.ovl16.gz — the overlap file of the grid case (two of every three records of reference `wtzmo -t 1` on the generated FASTA).
Fixtures are data (inputs + expected outputs); no reference source text is stored."""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gbo_inputs  # noqa: E402

GBO = os.path.join(ROOT, "oracle", "_ref", "wtgbo_ref")
ZMO = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")

# name, reads, overlap file (-j), extra argv  ('@x' = tests/golden/x, '@gen:grid4' = generated FASTA)
CASES = [
    ("tiny", "@tiny.fa.gz", "@zmo.ovl16.gz", []),
    ("tiny_opts", "@tiny.fa.gz", "@zmo.ovl16.gz", ["-Z", "50", "-R", "150", "-y", "600", "-w", "30", "-W", "1600", "-e", "400", "-M", "3", "-X", "-4", "-O", "-2", "-E", "-2",
                                                   "-T", "-20", "-l", "1", "-s", "300", "-m", "0.65", "-u", "200", "-q", "0.9", "-c", "0", "-N", "2"]),
    ("tiny_H_z12", "@tiny.fa.gz", "@zmo.ovl16.gz", ["-H", "-z", "12"]),
    ("tiny_Q", "@tiny.fa.gz", "@zmo.ovl16.gz", ["-Q", "-c", "2"]),
    ("tiny_b_L", "@tiny.fa.gz", "@zmo_b.ovl16.gz", ["-b", "@clips.txt", "-L", "@pairs.txt"]),
    ("tiny_dmo", "@tiny.fa.gz", "@dmo.ovl16.gz", ["-m", "0.1"]),
    ("tiny_n", "@tiny.fa.gz", "@zmo.ovl16.gz", ["-n"]),                    # kswx_refine_alignment behind the gates (hzm_aln.h:1721-1729)
    ("grid4_n_w20", "@gen:grid4", "@grid4.ovl16.gz", ["-n", "-w", "20", "-m", "0.7"]),
    ("grid4", "@gen:grid4", "@grid4.ovl16.gz", []),
    ("grid4_r300", "@gen:grid4", "@grid4.ovl16.gz", ["-r", "900", "-y", "500", "-R", "120"]),
]


def md5(b):
    return hashlib.md5(b).hexdigest()


def main():
    man = {"reference": "wtgbo -t 1 (oracle/_ref/wtgbo_ref)", "cases": {}, "generated": {}}
    with tempfile.TemporaryDirectory() as td:
        grid = os.path.join(td, "grid4.fa")
        man["generated"]["grid4"] = {"seed": 4, "md5": gbo_inputs.write_grid(grid, 4)}
        gj = os.path.join(HERE, "grid4.ovl16.gz")
        if not os.path.exists(gj):
            o = os.path.join(td, "grid4.zmo.ovl")
            subprocess.run([ZMO, "-t", "1", "-i", grid, "-fo", o, "-k", "16", "-s", "200", "-m", "0.6"], check=True, stderr=subprocess.DEVNULL)
            lines = [b"\t".join(l.split(b"\t")[:16]) for l in open(o, "rb").read().split(b"\n") if l]
            keep = [l for i, l in enumerate(lines) if (i + 1) % 3 != 0]
            with gzip.GzipFile(gj, "wb", mtime=0) as f:
                f.write(b"\n".join(keep) + b"\n")
        for name, reads, ovl, extra in CASES:
            rd = grid if reads == "@gen:grid4" else os.path.join(HERE, reads[1:])
            out = os.path.join(td, name + ".ovl"); pairs = os.path.join(td, name + ".pairs")
            argv = [a if not a.startswith("@") else os.path.join(HERE, a[1:]) for a in extra]
            subprocess.run([GBO, "-t", "1", "-i", rd, "-j", os.path.join(HERE, ovl[1:]), "-fo", out, "-9", pairs] + argv, check=True, stderr=subprocess.DEVNULL)
            full = open(out, "rb").read()
            lines = [l for l in full.split(b"\n") if l]
            cut = b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in lines) + (b"\n" if lines else b"")
            with gzip.GzipFile(os.path.join(HERE, "gbo_%s.ovl16.gz" % name), "wb", mtime=0) as f:
                f.write(cut)
            man["cases"][name] = {"reads": reads, "ovl": ovl, "argv": extra, "md5_full": md5(full), "md5_pairs": md5(open(pairs, "rb").read()),
                                  "records": len(lines), "repeated_lines": sum(1 for i in range(1, len(lines)) if lines[i] == lines[i - 1])}
            print(name, man["cases"][name]["records"], man["cases"][name]["repeated_lines"])
        # a read side beyond SG_MAX_EDGE: generated reads + a hand-written overlap file, one iteration (4-5 minutes of the reference)
        pf, po, ids = gbo_inputs.write_pile(td)
        out = os.path.join(td, "pile.ovl"); pairs = os.path.join(td, "pile.pairs")
        subprocess.run([GBO, "-t", "1", "-N", "1", "-i", pf, "-j", po, "-fo", out, "-9", pairs], check=True, stderr=subprocess.DEVNULL)
        full = open(out, "rb").read()
        man["edge_limit"] = dict(ids, argv=["-N", "1"], md5_full=md5(full), md5_pairs=md5(open(pairs, "rb").read()), records=full.count(b"\n"))
        print("edge_limit", man["edge_limit"]["records"])
    json.dump(man, open(os.path.join(HERE, "gbo_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
