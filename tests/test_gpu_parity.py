"""GPU parity tests (-m gpu): the drop-in `wtzmo` executable (host C -> C ABI -> HIP kernels on gfx950) must write
byte-identical .ovl / .contained files to the real reference `wtzmo -t 1` (committed goldens), to the oracle run live,
and - where the prebuilt reference binary travelled along - to the reference run live on a larger seeded input."""
import gzip
import os
import re
import subprocess

import pytest

from conftest import GOLD, ROOT, manifest, run_wtzmo_like
from smartdenovo_amd import synth

pytestmark = pytest.mark.gpu
CASES = sorted(manifest()["cases"].keys())
REF = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")


def test_library_is_the_hip_build(gpu_exe):
    from smartdenovo_amd import hipabi
    lib = hipabi.load()
    assert lib.wtz_device_count() >= 1, "no HIP device: these tests must run on an MI355X"


@pytest.mark.parametrize("name", CASES)
def test_gpu_equals_reference_golden(name, gpu_exe, tmp_path):
    case = manifest()["cases"][name]
    md5, cont, cut = run_wtzmo_like(gpu_exe, case, tmp_path, exact_pairs=True)
    assert cut == gzip.open(os.path.join(GOLD, name + ".ovl16.gz")).read(), "16-column records differ from the reference"
    assert md5 == case["md5_full"], "full .ovl (incl. CIGAR) differs from the reference"
    assert cont == case["md5_contained"]


@pytest.mark.parametrize("batch", ["1", "7", "4096"])
def test_batch_size_never_changes_the_output(batch, gpu_exe, tmp_path):
    """The speculative batch is an execution detail: any batch size must give the `-t 1` records."""
    case = manifest()["cases"]["zmo"]
    md5, cont, _ = run_wtzmo_like(gpu_exe, case, tmp_path, extra=["--batch", batch])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name", ["zmo", "dmo"])
def test_two_worker_contexts_same_output(name, gpu_exe, tmp_path):
    """--workers 2: wtz_ctx_clone, two HIP streams / scratch pools, batches committed strictly in sequence."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(gpu_exe, case, tmp_path, extra=["--workers", "2", "--batch", "8"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name,devs", [("zmo", "0,0"), ("dmo", "0,0,0"), ("zmo_S1", "0,0,0")])
def test_index_sharded_by_read_id_equals_unsharded(name, devs, gpu_exe, tmp_path):
    """--shard-index: contexts on this box's one GPU stand in for the devices; each indexes one read-id range, the k-mer filter uses the
    counts of all shards, every query is answered by every shard: the UNSHARDED reference records must come out (SURVEY 8e, configs[4])."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(gpu_exe, case, tmp_path, extra=["--gpu-list", devs, "--shard-index", "--batch", "16"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name,devs", [("zmo", "0,0"), ("dmo", "0,0"), ("zmo_n", "0,0,0")])
def test_multi_device_central_commit(name, devs, gpu_exe, tmp_path):
    """--gpu-list 0,0: two (three) contexts on this box's one GPU stand in for --gpus N: pairs dealt round-robin to the contexts, both
    indexes built per context, ONE in-order commit -> the plain `wtzmo -t 1` golden, not the union of -P stripes."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(gpu_exe, case, tmp_path, extra=["--gpu-list", devs, "--pool-mb", "8192"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


def test_scratch_pool_exhaustion_is_survived_or_loud(gpu_exe, tmp_path):
    """WTZ_E_POOL on the device: a batch that does not fit is halved (same output); a pool too small for one query is a loud
    exit(1) - never a memory fault, never a wrong file."""
    import hashlib
    case = manifest()["cases"]["zmo"]
    split_ok = planned = 0
    for mb in (4, 16, 48, 96, 160, 256, 512):
        out = os.path.join(str(tmp_path), "o%d.ovl" % mb)
        r = subprocess.run([gpu_exe, "-i", os.path.join(GOLD, case["input"]), "-fo", out, "--pool-mb", str(mb)] + case["argv"], capture_output=True)
        if r.returncode == 0:
            assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["md5_full"], "pool %d MB: wrong output" % mb
            split_ok += b"splitting the batch" in r.stderr
            m = re.search(rb"(\d+) batches in (\d+) ranges", r.stderr)
            planned += bool(m) and int(m.group(2)) > int(m.group(1))       # the batch was cut to the pool BEFORE the device stages ran
        else:
            assert r.returncode == 1 and b"scratch pool" in r.stderr, "pool %d MB: rc %d, %s" % (mb, r.returncode, r.stderr.decode()[-500:])
    assert split_ok >= 1 or planned >= 1, "no pool size exercised the planned ranges / the batch-splitting path"


@pytest.mark.parametrize("engine", ["zmo", "dmo"])
def test_injected_pool_failures_never_fault(engine, gpu_exe, oracle_exe, tmp_path):
    """Fault injection into the device scratch pools (the n-th request of one stage call fails as if the pool were full, at points a
    real exhaustion reaches only with the right mix of pairs): the run must finish with the reference's records after the driver's
    retry, or stop loudly - never a GPU memory fault or a hang.  (A window counted before its failed push used to index wins.a[-1]:
    the memory fault of the repeat-rich set.)"""
    names, seqs = synth.synth_reads(150000, 10, seed=123, mean_len=9000.0, min_len=1000, repeats=True)
    fa = os.path.join(str(tmp_path), "r.fa")
    synth.write_fasta(fa, names, seqs)
    argv = {"zmo": ["-k", "16", "-s", "200", "-m", "0.6"], "dmo": ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"]}[engine]
    ref = os.path.join(str(tmp_path), "ora.ovl")
    subprocess.run([oracle_exe, "-i", fa, "-fo", ref] + argv, check=True, capture_output=True)
    want = open(ref, "rb").read()
    ok = retried = 0
    plans = [("WTZ_POOL_FAIL_AT", n) for n in (2, 30, 300, 2500, 6000, 20000)] + ([("WTZ_TPOOL_FAIL_AT", n) for n in (1, 40, 900)] if engine == "zmo" else [])
    for var, n in plans:
        out = os.path.join(str(tmp_path), "o.ovl")
        env = dict(os.environ, WTZ_POOL_FAIL_ONCE="1")
        env[var] = str(n)
        r = subprocess.run([gpu_exe, "--pool-mb", "8192", "-i", fa, "-fo", out] + argv, capture_output=True, env=env, timeout=300)
        assert r.returncode in (0, 1), "%s=%d: rc %d: %s" % (var, n, r.returncode, r.stderr.decode()[-600:])
        if r.returncode == 0:
            assert open(out, "rb").read() == want, "%s=%d: wrong records after the retry" % (var, n)
            ok += 1
            retried += b"splitting the batch" in r.stderr
        else:
            assert b"scratch" in r.stderr or b"pool" in r.stderr
    assert ok >= 3 and retried >= 1


FRESH = {
    "zmo": ["-k", "16", "-s", "200", "-m", "0.6"],
    "dmo": ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"],
    # wide K-sw1 / K-sw2 bands: 4 and 8 band columns per lane in the register DPs, trace in the pool
    "zmo_w100": ["-k", "16", "-s", "200", "-m", "0.6", "-w", "100"],
    "zmo_w200": ["-k", "16", "-s", "200", "-m", "0.6", "-w", "200", "-W", "800"],
}


@pytest.mark.parametrize("engine", list(FRESH))
def test_gpu_equals_oracle_on_fresh_input(engine, gpu_exe, oracle_exe, tmp_path):
    names, seqs = synth.synth_reads(300000, 12, seed=99, mean_len=9000.0, min_len=1000)
    fa = os.path.join(str(tmp_path), "r.fa")
    synth.write_fasta(fa, names, seqs)
    argv = FRESH[engine]
    a, b = os.path.join(str(tmp_path), "gpu.ovl"), os.path.join(str(tmp_path), "ora.ovl")
    subprocess.run([gpu_exe, "-i", fa, "-fo", a] + argv, check=True, capture_output=True)
    subprocess.run([oracle_exe, "-i", fa, "-fo", b] + argv, check=True, capture_output=True)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a + ".contained", "rb").read() == open(b + ".contained", "rb").read()
    if os.path.exists(REF):
        c = os.path.join(str(tmp_path), "ref.ovl")
        subprocess.run([REF, "-t", "1", "-i", fa, "-fo", c] + argv, check=True, capture_output=True)
        assert open(a, "rb").read() == open(c, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["zmo", "dmo"])
def test_gpu_equals_oracle_on_repeat_rich_input(engine, gpu_exe, oracle_exe, tmp_path):
    """Tandem arrays + dispersed repeats: window scans with thousands of matches, gaps whose band doubles, large strand images (dmo tiers)."""
    names, seqs = synth.synth_reads(150000, 10, seed=123, mean_len=9000.0, min_len=1000, repeats=True)
    fa = os.path.join(str(tmp_path), "r.fa")
    synth.write_fasta(fa, names, seqs)
    argv = FRESH[engine]
    a, b = os.path.join(str(tmp_path), "gpu.ovl"), os.path.join(str(tmp_path), "ora.ovl")
    subprocess.run([gpu_exe, "-i", fa, "-fo", a] + argv, check=True, capture_output=True)
    subprocess.run([oracle_exe, "-i", fa, "-fo", b] + argv, check=True, capture_output=True)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a + ".contained", "rb").read() == open(b + ".contained", "rb").read()


FORMS = [
    {"WTZ_WINALIGN_LANE": "0"},                       # K-sw1: every window on the chained wave kernel (round-2 form)
    {"WTZ_WINALIGN_LANE": "2"},                       # both K-sw1 paths, every window compared on the device (fails loudly on a difference)
    {"WTZ_GAP_LANE": "0"},                            # K-sw2: every gap on a wavefront
    {"WTZ_CAND_WG": "0"},                             # seed lookup: wave per query, whole-query sort
    {"WTZ_CAND_STREAM": "1", "WTZ_CAND_WG": "0"},     # seed lookup: sort-free accumulation
    {"WTZ_RANGE_OVERLAP": "0"},                       # host: strict plan -> compute -> commit order
    {"WTZ_BATCH_OVERLAP": "0"},                       # host: one batch at a time (ranges still pipelined)
    {"WTZ_BATCH_OVERLAP_GAIN": "1e9"},              # host: every batch formed in front of the commit before it, whatever the mask rate
    {"WTZ_EXT_MW_ROWS": "600"},                       # K-sw3: the items whose extensions can run >= 600 rows on FOUR wavefronts each (frame form, wtz_sw_frame_mw.h) beside the fused launch
    {"WTZ_XCD_GROUP": "0"},                           # K_pair: identity block -> pair mapping
    {"WTZ_GAP_SIDESTREAM": "1"},                      # gaps on a side stream beside the left extensions
    {"WTZ_WINALIGN4": "1"},                           # four windows per wavefront
    {"WTZ_SW_CHECK": "1"},                            # scalar body beside every wave DP
    {"WTZ_EXT_PK": "0"},                              # K-sw3: the 32-bit frame form alone (round 5-6), without the packed 16-bit form in front
    {"WTZ_EXT_FUSED": "0"},                           # K-sw3: the two end extensions in two launches with K_stitch_mid between them (rounds 1-4) instead of on one wavefront
    {"WTZ_ZREAD": "0"},                               # z-mer index: device-wide fill + radix sort for every read instead of one workgroup per read
    {"WTZ_ZREAD": "0", "WTZ_ZCHUNK_M": "0.02"},       # ... built in many small chunks of reads
    {"WTZ_RANGE_FILL": "0.5", "WTZ_GAP_SIDESTREAM": "1"},   # smaller ranges; gaps on the side stream in front of the fused extension launch
]


@pytest.fixture(scope="module")
def forms_input(oracle_exe, tmp_path_factory):
    d = str(tmp_path_factory.mktemp("forms"))
    names, seqs = synth.synth_reads(200000, 12, seed=314, mean_len=9000.0, min_len=1000)
    fa = os.path.join(d, "r.fa")
    synth.write_fasta(fa, names, seqs)
    b = os.path.join(d, "ora.ovl")
    subprocess.run([oracle_exe, "-i", fa, "-fo", b] + FRESH["zmo"], check=True, capture_output=True)
    return fa, open(b, "rb").read(), open(b + ".contained", "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("env", FORMS, ids=["_".join("%s%s" % (k[4:], v) for k, v in sorted(e.items())) for e in FORMS])
def test_every_switchable_form_writes_the_same_records(env, gpu_exe, forms_input, tmp_path):
    """The library ships alternative forms of several stages behind environment switches (the previous round's kernels, experiments that did
    not pay, cross-checks).  None may change the output: each runs a fresh seeded read set with enough reads for every stage to have work
    and must write the oracle's (= the reference's) bytes."""
    fa, want, want_contained = forms_input
    a = os.path.join(str(tmp_path), "gpu.ovl")
    r = subprocess.run([gpu_exe, "-i", fa, "-fo", a, "--batch", "64"] + FRESH["zmo"], capture_output=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(want) > 10000
    assert open(a, "rb").read() == want, "%r changed the records" % (env,)
    assert open(a + ".contained", "rb").read() == want_contained
