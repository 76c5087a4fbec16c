#!/usr/bin/env python3
"""Parity pin for BASELINE configs[3] (fly-shape, ~10 Gbp of reads): the REAL reference `wtzmo -t 1 -P <n> -p 0` - one stripe of the queries
against the FULL index (wtzmo.c:1291,1314) - on the seeded synthetic set G = 140 Mbp x 70 (SURVEY 8d "fly-shape").  The whole job at -t 1
would take days here; a stripe takes about an hour and pins the same code path (index over all reads, candidate search, pair stages, commit).
Only checksums are committed (tests/golden/big_manifest.json, case "fly70_zmo_P<n>p0"); the GPU test regenerates the reads on the box.

Run in the build container only (oracle/_ref/wtzmo_ref):   python tests/golden/make_fly_stripe.py [--jobs-total 128] [--coverage 70]
"""
import argparse, hashlib, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
from smartdenovo_amd import synth  # noqa: E402
REF = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref"); MAN = os.path.join(HERE, "big_manifest.json")
ZMO = ["-k", "16", "-s", "200", "-m", "0.6"]


def file_md5(path):
    h = hashlib.md5(); n = 0
    with open(path, "rb") as fh:
        while True:
            b = fh.read(1 << 24)
            if not b:
                break
            h.update(b); n += b.count(b"\n")
    return h.hexdigest(), n


def main(shape="fly", genome=140000000, coverage=70.0, seed=53, jobs_total=128, zmo=ZMO, tmp="/tmp/wtz_fly"):
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=genome); ap.add_argument("--coverage", type=float, default=coverage); ap.add_argument("--seed", type=int, default=seed)
    ap.add_argument("--jobs-total", type=int, default=jobs_total); ap.add_argument("--tmp", default=tmp)
    a = ap.parse_args()
    os.makedirs(a.tmp, exist_ok=True)
    fa = os.path.join(a.tmp, "%s_G%d_c%g_s%d.fa" % (shape, a.genome, a.coverage, a.seed))
    t0 = time.time()
    if not os.path.exists(fa + ".meta"):
        names, seqs = synth.synth_reads(a.genome, a.coverage, seed=a.seed)
        md5 = synth.write_fasta(fa, names, seqs)
        json.dump({"reads": len(names), "bases": int(sum(s.size for s in seqs)), "md5": md5}, open(fa + ".meta", "w"))
        del names, seqs
    meta = json.load(open(fa + ".meta"))
    print("reads:", meta, "%.0f s" % (time.time() - t0), flush=True)
    name = "%s%g_zmo_P%dp0" % (shape, a.coverage, a.jobs_total)
    out = os.path.join(a.tmp, name + ".ovl"); pairs = os.path.join(a.tmp, name + ".pairs")
    t1 = time.time()
    subprocess.run([REF, "-t", "1", "-P", str(a.jobs_total), "-p", "0", "-i", fa, "-fo", out, "-9", pairs] + zmo, check=True, stdout=subprocess.DEVNULL)
    dt = time.time() - t1
    md5, nrec = file_md5(out)
    cont, ncont = file_md5(out + ".contained")
    lens = {}
    nm = None
    for line in open(fa):
        if line[0] == ">":
            nm = line[1:].strip()
        else:
            lens[nm] = len(line.strip())
    npair = 0; bp = 0
    for line in open(pairs):
        x, y = line.split(); npair += 1; bp += lens[x] + lens[y]
    man = json.load(open(MAN))
    sname = "%s%g" % (shape, a.coverage)
    man["sets"][sname] = dict(genome=a.genome, coverage=a.coverage, seed=a.seed, repeats=False, reads=meta["reads"], bases=meta["bases"], md5_fasta=meta["md5"])
    man["cases"][name] = dict(set=sname, engine="zmo", argv=list(zmo) + ["-P", str(a.jobs_total), "-p", "0"], md5_full=md5, records=nrec, md5_contained=cont, contained=ncont,
                              pairs=npair, pair_bp=bp, reference_seconds=round(dt, 1))
    json.dump(man, open(MAN, "w"), indent=1, sort_keys=True)
    print(name, man["cases"][name], flush=True)


if __name__ == "__main__":
    main()
