#!/usr/bin/env python3
"""BASELINE-scale goldens: md5 / record count of the REAL reference `wtzmo -t 1` on the seeded synthetic read sets the
bench and the large GPU parity tests regenerate with smartdenovo_amd/synth.py (same seed -> same bytes; the md5 of the
FASTA is recorded so generator drift is detected).

Run in the build container only (needs oracle/_ref/wtzmo_ref = /root/reference compiled where it lies by oracle/Makefile).
Only checksums are committed (tests/golden/big_manifest.json): the .ovl files are 0.2 - 3 GB.

  python tests/golden/make_big_goldens.py [set ...]        sets: ecoli yeast30 yeast100 repeat (default: all)

Each (set, engine) is one reference process; they run concurrently (--jobs).  The manifest is updated after every
finished job, so the script can be interrupted and re-run (finished cases are skipped unless --force).
"""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from smartdenovo_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")
MAN = os.path.join(HERE, "big_manifest.json")
ENG = {
    "zmo": ["-k", "16", "-s", "200", "-m", "0.6"],                                           # smartdenovo.pl:57-58
    "dmo": ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"],      # smartdenovo.pl:47-48
}
# name -> synth_reads(genome, coverage, seed, repeats); shapes of BASELINE.json configs[1] / configs[2]
SETS = {
    "ecoli": dict(genome=4600000, coverage=25.0, seed=11, repeats=False),       # configs[1]: E. coli-shape, 115 Mbp of reads
    "yeast30": dict(genome=12000000, coverage=30.0, seed=23, repeats=False),    # yeast genome size, 360 Mbp of reads
    "yeast100": dict(genome=12000000, coverage=100.0, seed=29, repeats=False),  # configs[2]: 1.2 Gbp of reads
    "repeat": dict(genome=2000000, coverage=20.0, seed=41, repeats=True),       # tandem arrays + dispersed copies, 40 Mbp
}
LOCK = threading.Lock()


def file_md5(path):
    h = hashlib.md5()
    n = 0
    with open(path, "rb") as fh:
        while True:
            b = fh.read(1 << 24)
            if not b:
                break
            h.update(b)
            n += b.count(b"\n")
    return h.hexdigest(), n


def load():
    return json.load(open(MAN)) if os.path.exists(MAN) else {"reference": "ruanjue/smartdenovo wtzmo, built by oracle/Makefile, run with -t 1", "sets": {}, "cases": {}}


def save(man):
    json.dump(man, open(MAN + ".tmp", "w"), indent=1, sort_keys=True)
    os.replace(MAN + ".tmp", MAN)


def gen_input(name, tmp):
    p = SETS[name]
    fa = os.path.join(tmp, "big_%s.fa" % name)
    if os.path.exists(fa) and os.path.exists(fa + ".meta"):
        return fa, json.load(open(fa + ".meta"))
    names, seqs = synth.synth_reads(p["genome"], p["coverage"], seed=p["seed"], repeats=p["repeats"])
    md5 = synth.write_fasta(fa + ".tmp", names, seqs)
    os.replace(fa + ".tmp", fa)
    meta = dict(p, reads=len(names), bases=int(sum(s.size for s in seqs)), md5_fasta=md5)
    json.dump(meta, open(fa + ".meta", "w"))
    return fa, meta


def run_case(name, eng, fa, tmp):
    out = os.path.join(tmp, "big_%s.%s.ref.ovl" % (name, eng))
    pairs = out + ".pairs"
    t0 = time.time()
    subprocess.run([REF, "-t", "1", "-i", fa, "-fo", out, "-9", pairs] + ENG[eng], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    md5, nrec = file_md5(out)
    cmd5, ncont = file_md5(out + ".contained")
    lens = {}
    nm = None
    for line in open(fa):
        if line[0] == ">":
            nm = line[1:].strip()
        else:
            lens[nm] = len(line) - 1
    bp = npair = 0
    for line in open(pairs):
        a, b = line.split()
        bp += lens[a] + lens[b]
        npair += 1
    res = {"set": name, "engine": eng, "argv": ENG[eng], "md5_full": md5, "records": nrec, "md5_contained": cmd5, "contained": ncont,
           "pairs": npair, "pair_bp": bp, "ovl_bytes": os.path.getsize(out), "ref_t1_seconds": round(dt, 1)}
    for f in (out, pairs, out + ".contained"):
        os.remove(f)
    with LOCK:
        man = load()
        man["cases"]["%s_%s" % (name, eng)] = res
        save(man)
    print("done %s_%s: %d records in %.0f s" % (name, eng, nrec, dt), flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sets", nargs="*", default=list(SETS))
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--tmp", default="/tmp/wtz_big")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.tmp, exist_ok=True)
    todo = []
    for name in a.sets:
        fa, meta = gen_input(name, a.tmp)
        with LOCK:
            man = load()
            man["sets"][name] = meta
            save(man)
        print("input %s: %d reads, %d bases, md5 %s" % (name, meta["reads"], meta["bases"], meta["md5_fasta"]), flush=True)
        for eng in ENG:
            if a.force or ("%s_%s" % (name, eng)) not in load()["cases"]:
                todo.append((name, eng, fa))
    # longest first
    todo.sort(key=lambda t: -SETS[t[0]]["genome"] * SETS[t[0]]["coverage"] ** 2)
    with cf.ThreadPoolExecutor(a.jobs) as ex:
        for f in [ex.submit(run_case, n, e, fa, a.tmp) for n, e, fa in todo]:
            f.result()


if __name__ == "__main__":
    main()
