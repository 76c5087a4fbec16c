"""The oracle (oracle/, CPU restatement) must reproduce, byte for byte, every golden fixture that the REAL
reference `wtzmo -t 1` produced (tests/golden/make_goldens.py): full .ovl incl. CIGAR (md5), .contained, and the
committed 16-column text.  This is what pins the oracle (the reference itself has no tests)."""
import gzip
import os

import pytest

from conftest import GOLD, manifest, run_wtzmo_like

CASES = sorted(manifest()["cases"].keys())


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_golden(name, oracle_exe, tmp_path):
    case = manifest()["cases"][name]
    md5, cont, cut = run_wtzmo_like(oracle_exe, case, tmp_path)
    assert cut == gzip.open(os.path.join(GOLD, name + ".ovl16.gz")).read(), "16-column records differ from the reference"
    assert md5 == case["md5_full"], "full .ovl (incl. CIGAR) differs from the reference"
    assert cont == case["md5_contained"]


def test_C_switch_is_a_noop_on_records():
    """-C never reaches wt->skip_contained in the reference (wtzmo.c:1609 vs 168): identical records, no side file."""
    m = manifest()["cases"]
    assert m["zmo"]["md5_full"] == m["zmo_C"]["md5_full"] and m["zmo_C"]["md5_contained"] is None
