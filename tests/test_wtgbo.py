"""f1: the `wtgbo` drop-in.  Goldens are outputs of the REAL reference `wtgbo -t 1` (tests/golden/make_gbo_goldens.py).

CPU: the host driver (overlap graph, candidate walks, commit order incl. the reference's repeated-line quirk, -9 slot order) on the
emulated device layer must write the reference's bytes.  GPU (-m gpu): the same through bin/wtgbo -> C-ABI -> HIP kernels, plus a
fresh 3 600-read input against the reference run live (oracle/_ref travels to the GPU box)."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLD, ROOT, md5_file
import gbo_inputs

MAN = json.load(open(os.path.join(GOLD, "gbo_manifest.json")))
CASES = sorted(MAN["cases"])
REF_GBO = os.path.join(ROOT, "oracle", "_ref", "wtgbo_ref")
REF_ZMO = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")


def _reads(case, tmp):
    if case["reads"].startswith("@gen:"):
        g = MAN["generated"][case["reads"][5:]]
        p = os.path.join(str(tmp), case["reads"][5:] + ".fa")
        if not os.path.exists(p):
            gbo_inputs.write_grid(p, g["seed"], g["md5"])
        return p
    return os.path.join(GOLD, case["reads"][1:])


def run_gbo(exe, case, tmp, extra=()):
    out = os.path.join(str(tmp), "g.ovl"); pairs = os.path.join(str(tmp), "g.pairs")
    argv = [a if not a.startswith("@") else os.path.join(GOLD, a[1:]) for a in case["argv"]]
    cmd = [exe, "-t", "1", "-i", _reads(case, tmp), "-j", os.path.join(GOLD, case["ovl"][1:]), "-fo", out, "-9", pairs] + argv + list(extra)
    r = subprocess.run(cmd, capture_output=True)
    assert r.returncode == 0, "%s failed (%d): %s" % (" ".join(cmd), r.returncode, r.stderr.decode()[-2000:])
    return md5_file(out), md5_file(pairs), open(out, "rb").read(), r.stderr.decode()


def check_case(exe, name, tmp, extra=()):
    case = MAN["cases"][name]
    m, mp, full, err = run_gbo(exe, case, tmp, extra)
    if m != case["md5_full"]:       # say where
        want = gzip.open(os.path.join(GOLD, "gbo_%s.ovl16.gz" % name)).read().split(b"\n")
        got = [b"\t".join(l.split(b"\t")[:16]) for l in full.split(b"\n")]
        for i, (a, b) in enumerate(zip(want, got)):
            assert a == b, "record %d differs:\n ref %s\n got %s" % (i, a.decode(), b.decode())
        assert len(want) == len(got), "record count %d != %d" % (len(got), len(want))
    assert m == case["md5_full"], "17-column output differs from the reference (CIGAR column)"
    assert mp == case["md5_pairs"], "-9 pair file differs from the reference's byte for byte"
    return err


@pytest.fixture(scope="module")
def emul_gbo():
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return os.path.join(ROOT, "tests", "emul", "wtgbo_emul")


@pytest.fixture(scope="module")
def oracle_gbo():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "wtgbo_oracle"], check=True)
    return os.path.join(ROOT, "oracle", "wtgbo_oracle")


@pytest.mark.parametrize("name", CASES)
def test_wtgbo_oracle_equals_reference_golden(name, oracle_gbo, tmp_path):
    """pins oracle/ora_hzmaux.h (align_hzmaux on the CPU) at program level"""
    check_case(oracle_gbo, name, tmp_path)


@pytest.mark.parametrize("name", CASES)
def test_wtgbo_host_logic_on_emulated_device(name, emul_gbo, tmp_path):
    check_case(emul_gbo, name, tmp_path)


def test_the_repeated_line_quirk_is_in_the_goldens():
    """wtgbo.c:185-191: a hit that covers the whole candidate ends the node's list and is printed twice; the grid case must exercise it"""
    assert MAN["cases"]["grid4"]["repeated_lines"] >= 1


@pytest.mark.parametrize("extra", [["--batch", "7"], ["--batch", "1"], ["--zindex-batch", "1", "--batch", "64"]])
def test_wtgbo_batching_never_changes_the_output(extra, emul_gbo, tmp_path):
    err = check_case(emul_gbo, "grid4", tmp_path, extra)
    assert "dropped behind a containing hit" in err


def test_wtgbo_cli_errors_like_the_reference(emul_gbo, tmp_path):
    """wtgbo.c:453-459: missing -o / -i / -j or an existing output without -f -> usage on STDOUT, return 1"""
    r = subprocess.run([emul_gbo], capture_output=True)
    assert r.returncode == 1 and b"Usage: wtgbo" in r.stdout
    out = os.path.join(str(tmp_path), "x.ovl"); open(out, "w").write("keep\n")
    r = subprocess.run([emul_gbo, "-i", os.path.join(GOLD, "tiny.fa.gz"), "-j", os.path.join(GOLD, "zmo.ovl16.gz"), "-o", out], capture_output=True)
    assert r.returncode == 1 and b"File exists" in r.stderr and open(out).read() == "keep\n"


def test_wtgbo_output_to_stdout(emul_gbo, tmp_path):
    """the pipeline's form: `wtgbo ... -fo - | cut -f1-16` (smartdenovo.pl:61)"""
    case = MAN["cases"]["tiny"]
    r = subprocess.run([emul_gbo, "-t", "4", "-i", os.path.join(GOLD, "tiny.fa.gz"), "-j", os.path.join(GOLD, "zmo.ovl16.gz"), "-fo", "-"], capture_output=True)
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == case["md5_full"]


@pytest.mark.skipif(not (os.path.exists(REF_GBO) and os.path.exists(REF_ZMO)), reason="reference binaries not built (make -C oracle ref)")
def test_wtgbo_equals_live_reference_on_fresh_input(emul_gbo, tmp_path):
    from smartdenovo_amd import synth
    fa = os.path.join(str(tmp_path), "r.fa")
    names, seqs = synth.synth_reads(400000, 18.0, seed=211)
    synth.write_fasta(fa, names, seqs)
    zo = os.path.join(str(tmp_path), "z.ovl")
    subprocess.run([REF_ZMO, "-t", "1", "-i", fa, "-fo", zo, "-k", "16", "-s", "200", "-m", "0.6"], check=True, stderr=subprocess.DEVNULL)
    z16 = os.path.join(str(tmp_path), "z.ovl16")
    open(z16, "wb").write(b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in open(zo, "rb").read().split(b"\n")))
    res = {}
    for tag, exe in (("ref", REF_GBO), ("emul", emul_gbo)):
        o = os.path.join(str(tmp_path), tag + ".ovl"); p = os.path.join(str(tmp_path), tag + ".pairs")
        subprocess.run([exe, "-t", "1", "-i", fa, "-j", z16, "-fo", o, "-9", p], check=True, stderr=subprocess.DEVNULL)
        res[tag] = (open(o, "rb").read(), open(p, "rb").read())
    assert len(res["ref"][0]) > 0
    assert res["ref"][0] == res["emul"][0] and res["ref"][1] == res["emul"][1]



def _run_edge_limit(exe, tmp):
    """a read side with 1 100 overlaps: the loader's 1 023-edge limit (SG_MAX_EDGE, wtlay.h:35, 459-463) decides which 522 753 pairs the anchoring pass aligns"""
    E = MAN["edge_limit"]
    pf, po, _ = gbo_inputs.write_pile(str(tmp), E)
    out = os.path.join(str(tmp), "p.ovl"); pairs = os.path.join(str(tmp), "p.pairs")
    r = subprocess.run([exe, "-t", "1", "-i", pf, "-j", po, "-fo", out, "-9", pairs] + E["argv"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    full = open(out, "rb").read()
    assert full.count(b"\n") == E["records"] == 1023 * 1022 // 2
    assert md5_file(out) == E["md5_full"], "output differs from the reference's at the 1 023-edge limit"
    assert md5_file(pairs) == E["md5_pairs"]


def test_wtgbo_edge_limit_of_a_read_side_equals_reference(oracle_gbo, tmp_path):
    _run_edge_limit(oracle_gbo, tmp_path)

# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu_gbo():
    import __graft_entry__ as ge
    if not (os.path.exists(ge.EXE_GBO) and os.path.exists(ge.LIB)):
        ge.build_product()
    return ge.EXE_GBO


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_wtgbo_equals_reference_golden(name, gpu_gbo, tmp_path):
    check_case(gpu_gbo, name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--batch", "50"], ["--zindex-batch", "1", "--batch", "300"]])
def test_gpu_wtgbo_batching_never_changes_the_output(extra, gpu_gbo, tmp_path):
    check_case(gpu_gbo, "grid4", tmp_path, extra)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_GBO) and os.path.exists(REF_ZMO)), reason="reference binaries did not travel")
def test_gpu_wtgbo_after_gpu_wtzmo_equals_reference_chain(gpu_gbo, gpu_exe, tmp_path):
    """the two pipeline steps back to back (smartdenovo.pl:57-61) on a 3 600-read set: bin/wtzmo | cut -f1-16 -> bin/wtgbo must equal
    reference wtzmo -t 1 | cut -> reference wtgbo -t 1 (the first step is bit-exact, so the second sees the same file)"""
    from smartdenovo_amd import synth
    fa = os.path.join(str(tmp_path), "r.fa")
    names, seqs = synth.synth_reads(1500000, 25.0, seed=7)
    synth.write_fasta(fa, names, seqs)
    res = {}
    for tag, zmo, gbo, t in (("gpu", gpu_exe, gpu_gbo, []), ("ref", REF_ZMO, REF_GBO, ["-t", "1"])):
        zo = os.path.join(str(tmp_path), tag + ".z.ovl")
        subprocess.run([zmo] + (["-t", "32"] if tag == "ref" else []) + ["-i", fa, "-fo", zo, "-k", "16", "-s", "200", "-m", "0.6"], check=True, stderr=subprocess.DEVNULL)
        res[tag] = zo
    # the reference step 1 ran with 32 workers (its output order then differs run to run); both second steps read the GPU's file, which equals `wtzmo -t 1`
    z16 = os.path.join(str(tmp_path), "z.ovl16")
    open(z16, "wb").write(b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in open(res["gpu"], "rb").read().split(b"\n")))
    outs = {}
    for tag, gbo in (("gpu", gpu_gbo), ("ref", REF_GBO)):
        o = os.path.join(str(tmp_path), tag + ".g.ovl"); p = os.path.join(str(tmp_path), tag + ".g.pairs")
        subprocess.run([gbo, "-t", "1", "-i", fa, "-j", z16, "-fo", o, "-9", p], check=True, stderr=subprocess.DEVNULL)
        outs[tag] = (open(o, "rb").read(), open(p, "rb").read())
    assert outs["ref"][0].count(b"\n") > 500
    assert outs["gpu"][0] == outs["ref"][0] and outs["gpu"][1] == outs["ref"][1]


@pytest.mark.gpu
def test_gpu_wtgbo_edge_limit_of_a_read_side_equals_reference(gpu_gbo, tmp_path):
    _run_edge_limit(gpu_gbo, tmp_path)
