import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
# A process that does not ask for a pool size gets the library's default: 45 % of the free HBM up to 128 GB = 4.4 s of hipMalloc.  The pool size never changes a
# result (tests/test_gpu_parity.py::test_scratch_pool_exhaustion_is_survived_or_loud, the --pool-mb cases), and a hundred small golden cases spent 400 of the GPU
# suite's 814 s allocating it (round 6, --durations).  The product's own default is still run by smoke(), bench.py and the scale tests' explicit sizes.
os.environ.setdefault("WTZ_DEFAULT_POOL_MB", "8192")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_present():
    """a ROCm device node this process can open: the `-m gpu` tests call the product, which has no CPU fallback"""
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing): gpu-marked tests run on the MI355X box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))


def case_argv(case, outdir=None):
    """'@name' = a side input under tests/golden/, '@out:name' = a side OUTPUT written next to the .ovl"""
    return [os.path.join(str(outdir or "."), a[5:]) if a.startswith("@out:") else (a if not a.startswith("@") else os.path.join(GOLD, a[1:])) for a in case["argv"]]


def pairs_md5(tmpdir):
    """md5 of the sorted lines of the -9 file (the reference writes it in hash-table order: the set is the contract)"""
    p = os.path.join(str(tmpdir), "pairs")
    if not os.path.exists(p):
        return None
    return hashlib.md5(b"\n".join(sorted(open(p, "rb").read().split(b"\n")))).hexdigest()


def md5_file(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None


def pairs_raw_md5(tmpdir):
    p = os.path.join(str(tmpdir), "pairs")
    return md5_file(p)


def run_wtzmo_like(exe, case, tmpdir, extra=(), exact_pairs=False):
    """Run a wtzmo-compatible executable on one golden case; returns (md5 of .ovl, md5 of .contained, 16-col text)."""
    out = os.path.join(str(tmpdir), "o.ovl")
    for f in (out, out + ".contained", os.path.join(str(tmpdir), "pairs")):
        if os.path.exists(f):
            os.remove(f)
    cmd = [exe, "-i", os.path.join(GOLD, case["input"]), "-fo", out] + case_argv(case, tmpdir) + list(extra)
    r = subprocess.run(cmd, capture_output=True)
    assert r.returncode == 0, "%s failed (%d): %s" % (" ".join(cmd), r.returncode, r.stderr.decode()[-2000:])
    full = open(out, "rb").read()
    cut = b"\n".join(b"\t".join(l.split(b"\t")[:16]) for l in full.split(b"\n"))
    assert pairs_md5(tmpdir) == case.get("md5_pairs_sorted"), "-9 pair set differs from the reference"
    if exact_pairs:       # the product also reproduces the ORDER (the oracle writes the set sorted)
        assert pairs_raw_md5(tmpdir) == case.get("md5_pairs"), "-9 pair file differs from the reference's byte for byte"
    return hashlib.md5(full).hexdigest(), md5_file(out + ".contained"), cut


@pytest.fixture(scope="session")
def oracle_exe():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "wtzmo_oracle"], check=True)
    return os.path.join(ROOT, "oracle", "wtzmo_oracle")


@pytest.fixture(scope="session")
def gpu_exe():
    import __graft_entry__ as ge
    if not (os.path.exists(ge.EXE) and os.path.exists(ge.LIB)):
        ge.build_product()
    return ge.EXE
