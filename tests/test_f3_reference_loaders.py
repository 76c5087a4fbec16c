"""f3, the part the row is named after: the REFERENCE's own two overlap loaders taking binary records -
`parse_overlap_item_strgraph` / `load_overlaps_strgraph` (wtlay.h:238-268, 443-470) and `load_alignments_wtclp` (wtclp.c:111-180).

Build container only (the reference sources must be present): integration/f3_patch_loaders.py copies the reference's C files to /tmp, inserts three lines
into wtlay.h and three into wtclp.c that call integration/wtz_ovlb_loaders.h, corrects the out-of-bounds read of wtclp.c:171-172 and compiles with plain
gcc.  The patched programs must write, from a binary stream (include/wtz_ovlb.h), exactly what they write from the text form of the same records -
`.lay` / `.clp` byte for byte - on the `tiny` golden, on the grid set of the wtgbo tests and on a 3 600-read set; the patched wtlay on TEXT must equal the
unpatched reference wtlay (oracle/_ref/wtlay_ref), so the inserted lines change nothing for text.  Nothing under tests/ holds reference source."""
import gzip
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import GOLD, ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are not on this machine (build container only)")


@pytest.fixture(scope="module")
def f3(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("f3"))
    subprocess.run([sys.executable, os.path.join(ROOT, "integration", "f3_patch_loaders.py"), "--ref", REF, "--out", out], check=True, capture_output=True)
    ovl = os.path.join(ROOT, "bin", "wtovl"); os.makedirs(os.path.dirname(ovl), exist_ok=True)
    subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", ovl, os.path.join(ROOT, "smartdenovo_amd", "csrc", "host", "wtovl_main.c")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/wtlay_ref", "_ref/wtzmo_ref"], check=True)
    return {"dir": out, "ovl": ovl, "lay_ref": os.path.join(ROOT, "oracle", "_ref", "wtlay_ref"), "zmo_ref": os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")}


def _inputs(which, f3, tmp):
    """(reads FASTA, 16-column text of its overlaps) - the tiny golden, or a fresh set overlapped by the reference itself"""
    if which == "tiny":
        txt = os.path.join(str(tmp), "tiny.ovl"); open(txt, "wb").write(gzip.open(os.path.join(GOLD, "zmo.ovl16.gz")).read())
        return os.path.join(GOLD, "tiny.fa.gz"), txt
    from smartdenovo_amd import synth
    names, seqs = synth.synth_reads(1200000, 30.0, seed=4242)           # 3 600 reads of ~10 kb, 15 % error
    fa = os.path.join(str(tmp), "set3600.fa"); synth.write_fasta(fa, names, seqs)
    full = os.path.join(str(tmp), "set3600.full.ovl"); txt = os.path.join(str(tmp), "set3600.ovl")
    subprocess.run([f3["zmo_ref"], "-t", "8", "-i", fa, "-fo", full, "-k", "16", "-s", "200", "-m", "0.6"], check=True, capture_output=True)
    with open(txt, "wb") as o:
        for l in open(full, "rb"):
            o.write(b"\t".join(l.rstrip(b"\n").split(b"\t")[:16]) + b"\n")
    assert len(names) > 3000 and os.path.getsize(txt) > 100000
    return fa, txt


def _md5s(prefix, exts):
    return {e: hashlib.md5(open(prefix + e, "rb").read()).hexdigest() for e in exts if os.path.exists(prefix + e)}


@pytest.mark.parametrize("which", ["tiny", "set3600"])
def test_reference_wtlay_and_wtclp_load_binary_records_like_text(which, f3, tmp_path):
    fa, txt = _inputs(which, f3, tmp_path)
    ovlb = os.path.join(str(tmp_path), which + ".ovlb")
    with open(ovlb, "wb") as o:
        subprocess.run([f3["ovl"], "-b", txt], check=True, stdout=o)
    back = subprocess.run([f3["ovl"], "-c", "16", ovlb], check=True, capture_output=True).stdout
    assert back == open(txt, "rb").read(), "text -> binary -> text is not the identity"
    assert os.path.getsize(ovlb) < os.path.getsize(txt)
    # ---- wtlay: patched on binary == patched on text == unpatched reference on text (every file it writes) ----
    lay = {}
    for tag, exe, ovl in (("ref_text", f3["lay_ref"], txt), ("bin_text", os.path.join(f3["dir"], "wtlay_bin"), txt), ("bin_binary", os.path.join(f3["dir"], "wtlay_bin"), ovlb)):
        pre = os.path.join(str(tmp_path), "lay_" + tag)
        r = subprocess.run([exe, "-i", fa, "-j", ovl, "-fo", pre + ".lay", "-s", "200", "-m", "0.6"], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lay[tag] = {f[len("lay_" + tag):]: hashlib.md5(open(os.path.join(str(tmp_path), f), "rb").read()).hexdigest() for f in sorted(os.listdir(str(tmp_path))) if f.startswith("lay_" + tag + ".")}
        assert ".lay" in lay[tag] and os.path.getsize(pre + ".lay") > 0
    assert lay["bin_text"] == lay["ref_text"], "the inserted lines changed wtlay's behaviour on text"
    assert lay["bin_binary"] == lay["ref_text"], "wtlay on binary records differs from wtlay on their text"
    # ---- wtclp: patched on binary == bounds-fixed reference on text ----
    clp = {}
    for tag, exe, ovl in (("fix_text", os.path.join(f3["dir"], "wtclp_fix"), txt), ("bin_text", os.path.join(f3["dir"], "wtclp_bin"), txt), ("bin_binary", os.path.join(f3["dir"], "wtclp_bin"), ovlb)):
        out = os.path.join(str(tmp_path), "clp_" + tag + ".clp")
        r = subprocess.run([exe, "-i", ovl, "-fo", out, "-d", "3", "-k", "300", "-m", "0.1", "-FT"], capture_output=True)
        assert r.returncode == 0, "%s rc %d: %s" % (tag, r.returncode, r.stderr.decode()[-1500:])
        clp[tag] = hashlib.md5(open(out, "rb").read()).hexdigest()
        assert os.path.getsize(out) > 0
    assert clp["bin_text"] == clp["fix_text"] and clp["bin_binary"] == clp["fix_text"], "wtclp on binary records differs from wtclp on their text: %r" % (clp,)


def test_unpatched_wtclp_reads_out_of_bounds_and_the_correction_is_minimal(f3, tmp_path):
    """what the correction is: ONE condition reordered (wtclp.c:171).  The unpatched loop reads ptrs[size] - one element past what was initialised - and indexes
    `hits` with it before it tests `i == size`; under a checking allocator that is a crash (SURVEY measured a segfault), under glibc's it depends on the heap."""
    a = open(os.path.join(REF, "wtclp.c")).read(); b = open(os.path.join(f3["dir"], "src", "wtclp_fix.c")).read()
    da = a.splitlines(); db = b.splitlines()
    assert len(da) == len(db)
    changed = [i for i, (x, y) in enumerate(zip(da, db)) if x != y]
    assert len(changed) == 1 and "i == wt->ptrs->size ||" in db[changed[0]] and db[changed[0]].index("i == wt->ptrs->size") < db[changed[0]].index("sids[1] =")


def test_reference_wtgbo_loads_binary_records_through_the_same_patch(f3, tmp_path):
    """the reference's wtgbo loads its -j files with wtlay.h's loader, so the patched header makes it binary-capable too: same records out as from text"""
    fa = os.path.join(GOLD, "tiny.fa.gz")
    txt = os.path.join(str(tmp_path), "tiny.ovl"); open(txt, "wb").write(gzip.open(os.path.join(GOLD, "zmo.ovl16.gz")).read())
    ovlb = os.path.join(str(tmp_path), "tiny.ovlb")
    with open(ovlb, "wb") as o:
        subprocess.run([f3["ovl"], "-b", txt], check=True, stdout=o)
    outs = []
    for ovl in (txt, ovlb):
        out = os.path.join(str(tmp_path), "g%d.ovl" % len(outs))
        subprocess.run([os.path.join(f3["dir"], "wtgbo_bin"), "-t", "1", "-i", fa, "-j", ovl, "-fo", out], check=True, capture_output=True)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 0
