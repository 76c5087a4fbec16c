"""The drop-in boundary at pipeline level: `smartdenovo.pl` writes a Makefile whose overlap rule calls $(EXE_ZMO); with EXE_ZMO = this
repo's `wtzmo` (here: the host driver on the emulated device layer - the same C main, argv handling and file output as bin/wtzmo) the rule
must produce, from the reads the reference's own `wtpre` prepared, exactly the file the reference `wtzmo -t 1` writes, and the reference's
`wtlay` must accept it.  Needs /root/reference (the generator script and the neighbour tools are run where they lie; nothing is copied):
skipped on the GPU box."""
import hashlib
import os
import subprocess

import pytest

from conftest import GOLD, ROOT

REFDIR = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFDIR), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("engine", ["dmo", "zmo"])
def test_generated_makefile_rule_runs_the_drop_in(engine, tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    ref = os.path.join(ROOT, "oracle", "_ref")
    bindir = os.path.join(str(tmp_path), "bin")
    os.makedirs(bindir)
    import __graft_entry__ as ge
    if not os.path.exists(ge.EXE_PRE):
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-o", os.path.join(str(tmp_path), "wtpre_built"),
                        os.path.join(ROOT, "smartdenovo_amd", "csrc", "host", "wtpre_main.c")], check=True)
    pre = ge.EXE_PRE if os.path.exists(ge.EXE_PRE) else os.path.join(str(tmp_path), "wtpre_built")
    # round 3: the steps either side of wtzmo are drop-ins too - wtpre (f4) in front, wtgbo (f1, on the emulated device layer here) behind
    links = {"wtpre": pre, "wtzmo": os.path.join(ROOT, "tests", "emul", "wtzmo_emul"), "wtlay": os.path.join(ref, "wtlay_ref"),
             "wtclp": "/bin/true", "wtobt": "/bin/true", "wtgbo": os.path.join(ROOT, "tests", "emul", "wtgbo_emul"), "wtcns": "/bin/true"}
    for n, t in links.items():
        os.symlink(t, os.path.join(bindir, n))
    env = dict(os.environ, PATH=bindir + ":" + os.environ["PATH"])
    reads = os.path.join(str(tmp_path), "reads.fa")
    subprocess.run("gzip -dc %s > %s" % (os.path.join(GOLD, "tiny.fa.gz"), reads), shell=True, check=True)
    mak = os.path.join(str(tmp_path), "asm.mak")
    with open(mak, "w") as fh:
        subprocess.run(["perl", os.path.join(REFDIR, "smartdenovo.pl"), "-p", "asm", "-e", engine, "-J", "3000", reads], stdout=fh, check=True, env=env, cwd=str(tmp_path))
    text = open(mak).read()
    assert "EXE_ZMO=" + os.path.join(bindir, "wtzmo") in text.replace(" ", ""), "smartdenovo.pl did not pick the drop-in up from PATH:\n" + text[:600]
    target = "asm.dmo.ovl" if engine == "dmo" else "asm.zmo.ovl.short"
    r = subprocess.run(["make", "-f", mak, target], cwd=str(tmp_path), env=env, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = open(os.path.join(str(tmp_path), target), "rb").read()
    assert got.count(b"\n") > 50
    # the prepared reads are what the reference's own wtpre writes for the rule `wtpre -J 3000 reads | gzip -c -1` (smartdenovo.pl:43-44)
    want_pre = subprocess.run([os.path.join(ref, "wtpre_ref"), "-J", "3000", reads], capture_output=True, check=True).stdout
    got_pre = subprocess.run(["gzip", "-dc", os.path.join(str(tmp_path), "asm.fa.gz")], capture_output=True, check=True).stdout
    assert got_pre == want_pre and got_pre.count(b">") > 100
    # what the reference's own wtzmo writes for the same prepared reads (same rule, -t 1)
    argv = ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"] if engine == "dmo" else ["-k", "16", "-s", "200", "-m", "0.6"]
    want_f = os.path.join(str(tmp_path), "ref.ovl")
    subprocess.run([os.path.join(ref, "wtzmo_ref"), "-t", "1", "-i", os.path.join(str(tmp_path), "asm.fa.gz"), "-fo", want_f] + argv, check=True, capture_output=True)
    want = open(want_f, "rb").read()
    if engine == "zmo":
        want = b"".join(b"\t".join(ln.split(b"\t")[:16]) + b"\n" for ln in want.split(b"\n") if ln)      # the rule pipes through cut -f1-16
    assert hashlib.md5(got).hexdigest() == hashlib.md5(want).hexdigest(), "the Makefile rule with the drop-in does not write the reference's records"
    if engine == "zmo":
        # the next rule of the zmo pipeline (smartdenovo.pl:60-61): $(EXE_GBO) ... -j asm.zmo.ovl.short -fo - | cut -f1-16 > asm.zmo.gbo.short
        r = subprocess.run(["make", "-f", mak, "asm.zmo.gbo.short"], cwd=str(tmp_path), env=env, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got_g = open(os.path.join(str(tmp_path), "asm.zmo.gbo.short"), "rb").read()
        rg = subprocess.run([os.path.join(ref, "wtgbo_ref"), "-t", "1", "-i", os.path.join(str(tmp_path), "asm.fa.gz"), "-j", os.path.join(str(tmp_path), target), "-fo", "-"],
                            capture_output=True, check=True, cwd=str(tmp_path)).stdout
        want_g = b"".join(b"\t".join(ln.split(b"\t")[:16]) + b"\n" for ln in rg.split(b"\n") if ln)
        assert got_g == want_g and got_g.count(b"\n") > 10, "the wtgbo rule with the drop-in does not write the reference's records"
    # the consumer: the reference's wtlay must load the overlaps and lay the reads out (wtlay.h:238-268 parses >= 16 columns)
    lay = os.path.join(str(tmp_path), "asm.lay")
    r = subprocess.run([os.path.join(bindir, "wtlay"), "-i", os.path.join(str(tmp_path), "asm.fa.gz"), "-j", os.path.join(str(tmp_path), target), "-fo", lay] +
                       (["-w", "300", "-s", "200", "-m", "0.1", "-r", "0.95", "-c", "1"] if engine == "dmo" else ["-s", "200", "-m", "0.6", "-R", "-r", "1", "-c", "1"]),
                       capture_output=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert os.path.getsize(lay) > 0 and open(lay).read().startswith(">"), "wtlay produced no layout from the drop-in's overlaps"
