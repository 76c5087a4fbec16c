"""f2: the `wtext` drop-in - overlaps clipped to the retained regions of their reads and extended to the region ends.
Goldens are outputs of the REAL reference `wtext -t 1` (tests/golden/make_ext_goldens.py).

CPU: (1) oracle/wtext_oracle (the product's host code over the oracle's shift-band extension) equals the goldens - that pins the host code;
(2) the same host code on the emulated device layer (wtz_extend_batch, every kernel a host loop) equals them for any device block size;
(3) live against oracle/_ref/wtext_ref on a fresh input where the reference binaries exist.
GPU (-m gpu): bin/wtext -> C-ABI -> the K-sw3 kernels, same goldens, plus the chain gpu wtzmo -> gpu wtext against the reference chain."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLD, ROOT, md5_file
import gbo_inputs

MAN = json.load(open(os.path.join(GOLD, "ext_manifest.json")))
CASES = sorted(MAN["cases"])
REF_EXT = os.path.join(ROOT, "oracle", "_ref", "wtext_ref")
REF_ZMO = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")
REF_OBT = os.path.join(ROOT, "oracle", "_ref", "wtobt_ref")
_made = {}


def _at(argv):
    return [a if not a.startswith("@") else os.path.join(GOLD, a[1:]) for a in argv]


def overlap_inputs(zmo_exe, tmp_factory, tag):
    """The 17-column inputs are regenerated (not stored); they are the reference's bytes or the test stops here."""
    key = (zmo_exe, tag)
    if key not in _made:
        d = tmp_factory.mktemp("ext_in_" + tag)
        fa = os.path.join(str(d), "tiny.fa")
        open(fa, "wb").write(gzip.open(os.path.join(GOLD, MAN["reads"])).read())
        files = {"fa": fa}
        for name, o in MAN["ovls"].items():
            out = os.path.join(str(d), name + ".ovl")
            r = subprocess.run([zmo_exe, "-i", fa, "-fo", out] + _at(o["argv"]), capture_output=True)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert md5_file(out) == o["md5_full"], "the regenerated overlap input %s is not the reference's" % name
            files[name] = out
        _made[key] = files
    return _made[key]


def run_ext(exe, name, files, tmp, extra=()):
    case = MAN["cases"][name]
    out = os.path.join(str(tmp), "x.ovl")
    js = []
    for o in case["ovls"]:
        js += ["-j", files[o]]
    cmd = [exe, "-t", "1", "-i", files["fa"]] + js + ["-fo", out] + _at(case["argv"]) + list(extra)
    r = subprocess.run(cmd, capture_output=True)
    assert r.returncode == 0, "%s failed (%d): %s" % (" ".join(cmd), r.returncode, r.stderr.decode()[-2000:])
    full = open(out, "rb").read()
    if md5_file(out) != case["md5_full"]:       # say where
        want = gzip.open(os.path.join(GOLD, "ext_%s.ovl16.gz" % name)).read().split(b"\n")
        got = [b"\t".join(l.split(b"\t")[:16]) for l in full.split(b"\n")]
        for i, (a, b) in enumerate(zip(want, got)):
            assert a == b, "record %d differs:\n ref %s\n got %s" % (i, a.decode(), b.decode())
        assert len(want) == len(got), "record count %d != %d" % (len(got), len(want))
    assert md5_file(out) == case["md5_full"], "17-column output differs from the reference (CIGAR column)"
    assert full.count(b"\n") == case["records"]
    return r.stderr.decode()


@pytest.fixture(scope="module")
def emul_ext():
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return os.path.join(ROOT, "tests", "emul", "wtext_emul")


@pytest.fixture(scope="module")
def oracle_ext():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "wtext_oracle"], check=True)
    return os.path.join(ROOT, "oracle", "wtext_oracle")


@pytest.mark.parametrize("name", CASES)
def test_wtext_oracle_equals_reference_golden(name, oracle_ext, oracle_exe, tmp_path_factory, tmp_path):
    run_ext(oracle_ext, name, overlap_inputs(oracle_exe, tmp_path_factory, "ora"), tmp_path)


@pytest.mark.parametrize("name", CASES)
def test_wtext_host_logic_on_emulated_device(name, emul_ext, oracle_exe, tmp_path_factory, tmp_path):
    err = run_ext(emul_ext, name, overlap_inputs(oracle_exe, tmp_path_factory, "ora"), tmp_path)
    assert "extension kernels" in err


@pytest.mark.parametrize("extra", [["--block", "1"], ["--block", "37"], ["--block", "100000", "-t", "8"]])
def test_wtext_blocks_never_change_the_output(extra, emul_ext, oracle_exe, tmp_path_factory, tmp_path):
    for name in ("hard", "brutal", "two_files"):
        run_ext(emul_ext, name, overlap_inputs(oracle_exe, tmp_path_factory, "ora"), tmp_path, extra)


def test_the_goldens_hold_the_cases_that_matter():
    c = MAN["cases"]
    assert c["brutal"]["records"] < c["none"]["records"]              # overlaps whose CIGAR does not survive the clipping are dropped (wtext.c:329)
    assert c["P2p0"]["records"] + c["P2p1"]["records"] == c["hard"]["records"] and c["P2p1"]["records"] == 100      # batches of 100 lines dealt round-robin
    assert c["prev"]["md5_full"] != c["none"]["md5_full"]


def test_wtext_cli_errors_like_the_reference(emul_ext, tmp_path):
    fa = os.path.join(str(tmp_path), "t.fa"); open(fa, "w").write(">a\nACGT\n")
    ovl = os.path.join(str(tmp_path), "t.ovl"); open(ovl, "w").write("")
    out = os.path.join(str(tmp_path), "o.ovl")
    for argv in (["-i", fa, "-j", ovl], ["-i", fa, "-o", out], ["-j", ovl, "-o", out], ["-h"], ["-S", "100", "-i", fa, "-j", ovl, "-o", out]):
        r = subprocess.run([emul_ext] + argv, capture_output=True)
        assert r.returncode == 1 and b"Usage: wtext" in r.stdout, argv          # usage on stdout, exit 1 (wtext.c:342-375); -S is not in the getopt string (wtext.c:406)
    open(out, "w").write("x")
    r = subprocess.run([emul_ext, "-i", fa, "-j", ovl, "-o", out], capture_output=True)
    assert r.returncode == 1 and b"File exists! '%s'" % out.encode() in r.stderr
    r = subprocess.run([emul_ext, "-i", fa, "-j", ovl, "-fo", out], capture_output=True)
    assert r.returncode == 0 and open(out).read() == ""


def test_wtext_output_to_stdout(emul_ext, oracle_exe, tmp_path_factory):
    f = overlap_inputs(oracle_exe, tmp_path_factory, "ora")
    r = subprocess.run([emul_ext, "-i", f["fa"], "-j", f["zmo"], "-b", os.path.join(GOLD, "ext_tiny.obt"), "-o", "-"], capture_output=True)
    import hashlib
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == MAN["cases"]["obt"]["md5_full"]


def _fresh_chain(zmo, obt, ext, fa, d, tag, zmo_extra=()):
    zo = os.path.join(d, tag + ".zmo.ovl"); ob = os.path.join(d, tag + ".obt"); xo = os.path.join(d, tag + ".ext.ovl")
    subprocess.run([zmo] + list(zmo_extra) + ["-i", fa, "-fo", zo, "-k", "16", "-s", "200", "-m", "0.6"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([obt, "-i", fa, "-j", zo, "-fo", ob, "-m", "0.6", "-c", "2"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([ext, "-t", "1", "-i", fa, "-j", zo, "-b", ob, "-fo", xo], check=True, stderr=subprocess.DEVNULL)
    return open(zo, "rb").read(), open(xo, "rb").read()


@pytest.mark.skipif(not all(os.path.exists(p) for p in (REF_EXT, REF_ZMO, REF_OBT)), reason="reference binaries not built (make -C oracle ref)")
def test_wtext_equals_live_reference_on_fresh_input(emul_ext, tmp_path):
    """wtzmo -> wtobt -> wtext as smartdenovo.pl's obt/ext steps chain them, on a generated read set (chimera-free, so wtobt mostly trims ends)."""
    fa = os.path.join(str(tmp_path), "grid9.fa"); gbo_inputs.write_grid(fa, 9)
    zr, xr = _fresh_chain(REF_ZMO, REF_OBT, REF_EXT, fa, str(tmp_path), "ref", ["-t", "8"])
    zo = os.path.join(str(tmp_path), "ref.zmo.ovl"); ob = os.path.join(str(tmp_path), "ref.obt"); xo = os.path.join(str(tmp_path), "emul.ext.ovl")
    subprocess.run([emul_ext, "-i", fa, "-j", zo, "-b", ob, "-fo", xo], check=True, stderr=subprocess.DEVNULL)
    assert xr.count(b"\n") > 500
    assert open(xo, "rb").read() == xr


# ---------------------------------------------------------------- GPU


@pytest.fixture(scope="module")
def gpu_ext(gpu_exe):
    import __graft_entry__ as ge
    if not os.path.exists(ge.EXE_EXT):
        ge.build_product()
    return ge.EXE_EXT


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_wtext_equals_reference_golden(name, gpu_ext, gpu_exe, tmp_path_factory, tmp_path):
    err = run_ext(gpu_ext, name, overlap_inputs(gpu_exe, tmp_path_factory, "gpu"), tmp_path)
    assert "extension kernels" in err


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--block", "1"], ["--block", "37"]])
def test_gpu_wtext_blocks_never_change_the_output(extra, gpu_ext, gpu_exe, tmp_path_factory, tmp_path):
    for name in ("hard", "two_files"):
        run_ext(gpu_ext, name, overlap_inputs(gpu_exe, tmp_path_factory, "gpu"), tmp_path, extra)


@pytest.mark.gpu
@pytest.mark.skipif(not all(os.path.exists(p) for p in (REF_EXT, REF_OBT)), reason="reference binaries not built (make -C oracle ref)")
def test_gpu_wtext_after_gpu_wtzmo_equals_reference_wtext(gpu_ext, gpu_exe, tmp_path):
    """~2 800 reads: gpu wtzmo -> reference wtobt -> gpu wtext against reference `wtext -t 1` on the same two files, byte for byte
    (bin/wtzmo's own parity with `wtzmo -t 1` is tests/test_gpu_parity.py's and test_gpu_scale.py's subject)."""
    d = str(tmp_path)
    fa = os.path.join(d, "grid21.fa")
    open(fa, "wb").write(gbo_inputs.grid_fasta(21, G=1200000))
    zo = os.path.join(d, "z.ovl"); ob = os.path.join(d, "z.obt"); xr = os.path.join(d, "ref.ext"); xg = os.path.join(d, "gpu.ext")
    subprocess.run([gpu_exe, "-i", fa, "-fo", zo, "-k", "16", "-s", "200", "-m", "0.6"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([REF_OBT, "-i", fa, "-j", zo, "-fo", ob, "-m", "0.6", "-c", "2"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([REF_EXT, "-t", "1", "-i", fa, "-j", zo, "-b", ob, "-fo", xr], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([gpu_ext, "-t", "8", "--block", "4096", "-i", fa, "-j", zo, "-b", ob, "-fo", xg], check=True, stderr=subprocess.DEVNULL)
    ref = open(xr, "rb").read()
    assert ref.count(b"\n") > 5000
    assert open(xg, "rb").read() == ref
