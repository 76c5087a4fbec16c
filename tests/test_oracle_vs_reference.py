"""Where the compiled reference is available (oracle/_ref/wtzmo_ref, built by `make -C oracle ref` from the sources
under /root/reference; the prebuilt binary also travels to the GPU box), run it live against the oracle on a fresh
seeded input that is NOT one of the committed fixtures."""
import os
import subprocess

import pytest

from conftest import ROOT
from smartdenovo_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (needs /root/reference once)")
@pytest.mark.parametrize("engine", ["zmo", "dmo"])
def test_oracle_equals_live_reference(engine, oracle_exe, tmp_path):
    names, seqs = synth.synth_reads(50000, 18, seed=2024, mean_len=6000.0, min_len=1000)
    fa = os.path.join(str(tmp_path), "r.fa")
    synth.write_fasta(fa, names, seqs)
    argv = ["-k", "16", "-s", "200", "-m", "0.6"] if engine == "zmo" else ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"]
    a, b = os.path.join(str(tmp_path), "ref.ovl"), os.path.join(str(tmp_path), "ora.ovl")
    subprocess.run([REF, "-t", "1", "-i", fa, "-fo", a] + argv, check=True, capture_output=True)
    subprocess.run([oracle_exe, "-i", fa, "-fo", b] + argv, check=True, capture_output=True)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a + ".contained", "rb").read() == open(b + ".contained", "rb").read()
    assert os.path.getsize(a) > 0
