"""CPU-side checks of the product: the C-ABI library exports every declared symbol (no compute call), the public
header and the ctypes mirror agree, and the host logic (batch planning + in-order commit of wtzmo_main.c) reproduces
the reference goldens when the device layer is emulated (tests/emul: every kernel run as a host loop)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import case_argv,  ROOT, manifest, run_wtzmo_like


def test_header_symbols_match_binding():
    from smartdenovo_amd import hipabi
    hdr = open(os.path.join(ROOT, "include", "wtzmo_hip.h")).read()
    declared = set(re.findall(r"\b(wtz_[a-z_]+)\s*\(", hdr))
    assert declared == set(hipabi.SYMBOLS)


def test_library_exports_all_symbols():
    from smartdenovo_amd import hipabi
    if not os.path.exists(hipabi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build_product()
    hipabi.check_symbols()


def test_params_struct_layout():
    from smartdenovo_amd import hipabi
    # 15 u32 + 2 f32 + 9 i32 + f32 + 5 i32 + 2 f32 + refine + aux_strand = 36 words
    assert ctypes.sizeof(hipabi.Params) == 36 * 4
    assert hipabi.PAIR_SUMMARY.itemsize == 48 and hipabi.WINBOX.itemsize == 16 and hipabi.ALN_RESULT.itemsize == 72


def test_no_gpu_means_loud_failure(tmp_path):
    """Without a HIP device the product must refuse to run (no CPU fallback)."""
    import __graft_entry__ as ge
    if not (os.path.exists(ge.EXE) and os.path.exists(ge.LIB)):
        ge.build_product()
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    case = manifest()["cases"]["zmo"]
    r = subprocess.run([ge.EXE, "-i", os.path.join(ROOT, "tests", "golden", case["input"]), "-fo", os.path.join(str(tmp_path), "x.ovl")] + case["argv"], capture_output=True)
    assert r.returncode != 0 and b"no HIP device" in r.stderr


@pytest.fixture(scope="module")
def emul_exe():
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return os.path.join(ROOT, "tests", "emul", "wtzmo_emul")


@pytest.mark.parametrize("name", ["zmo", "dmo", "zmo_G2", "zmo_N", "zmo_A5", "zmo_B2", "zmo_edge_fq", "zmo_L", "zmo_b", "dmo_U2", "zmo_P2p1", "zmo_n", "dmo_N", "zmo_I", "dmo_I", "zmo_9", "dmo_9"])
def test_host_logic_on_emulated_device(name, emul_exe, tmp_path):
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--batch", "16"], exact_pairs=True)
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name", ["zmo", "dmo", "zmo_B2", "zmo_L", "zmo_9", "zmo_n"])
@pytest.mark.parametrize("gain", ["1e9", "0"])
def test_batches_pipelined_or_one_at_a_time(name, gain, emul_exe, tmp_path, monkeypatch):
    """Round 5: the next batch is formed (from the prefetched candidates and the masks as they are) in front of the last commit of the batch before it and its first range
    runs meanwhile - where the work that commit is expected to mask costs less than the wait it hides.  Forced on at every boundary (WTZ_BATCH_OVERLAP_GAIN=1e9) and off everywhere (0) with
    batches of 7 queries: the same bytes as the reference, incl. the order of the -9 pair file."""
    monkeypatch.setenv("WTZ_BATCH_OVERLAP_GAIN", gain)
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--batch", "7"], exact_pairs=True)
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


def test_repeat_rich_window_scans_on_the_pool_workspace(emul_exe, oracle_exe, tmp_path):
    """Tandem arrays + dispersed repeats: thousands of matches of one strand inside one 800-column window - scans that do not fit the wave's LDS slice and run the
    wave-parallel body on a workspace in the pool (3 405 such scans on this input, up to 3 640 matches).  Output == the oracle's."""
    import hashlib
    from smartdenovo_amd import synth
    names, seqs = synth.synth_reads(60000, 8, seed=123, mean_len=9000.0, min_len=1000, repeats=True)
    fa = os.path.join(str(tmp_path), "rep.fa"); synth.write_fasta(fa, names, seqs)
    md5 = {}
    for tag, exe in (("emul", emul_exe), ("oracle", oracle_exe)):
        out = os.path.join(str(tmp_path), tag + ".ovl")
        r = subprocess.run([exe, "-i", fa, "-fo", out, "-k", "16", "-s", "200", "-m", "0.6"], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        md5[tag] = hashlib.md5(open(out, "rb").read()).hexdigest()
        assert open(out, "rb").read().count(b"\n") > 100
    assert md5["emul"] == md5["oracle"]


def test_scratch_pool_exhaustion_splits_the_batch(emul_exe, tmp_path):
    """WTZ_E_POOL path (wtzmo_main.c process_range): a pool too small for the whole batch halves it until it fits; the output
    is unchanged.  (A task that ran out of scratch once used to go on with a NULL row buffer: wtz_swmem_need is sticky now.)"""
    case = manifest()["cases"]["zmo"]
    out = os.path.join(str(tmp_path), "o.ovl")
    r = subprocess.run([emul_exe, "-i", os.path.join(ROOT, "tests", "golden", case["input"]), "-fo", out, "--pool-mb", "192"] + case["argv"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"splitting the batch" in r.stderr
    import hashlib
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["md5_full"]
    # far too small: a loud failure, never a crash or a wrong file
    r = subprocess.run([emul_exe, "-i", os.path.join(ROOT, "tests", "golden", case["input"]), "-fo", out, "--pool-mb", "8"] + case["argv"], capture_output=True)
    assert r.returncode == 1 and b"scratch pool" in r.stderr


@pytest.mark.parametrize("name", ["zmo", "dmo"])
def test_two_worker_contexts_same_output(name, emul_exe, tmp_path):
    """--workers 2: two batches in flight on two contexts, committed strictly in sequence (writer + commit are TSan-clean)."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--workers", "2", "--batch", "8"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name,devs", [("zmo", "0,0"), ("dmo", "0,0"), ("zmo", "0,0,0"), ("zmo_I", "0,0")])
def test_multi_device_central_commit(name, devs, emul_exe, tmp_path):
    """--gpus N / --gpu-list: one process, one context per device, the pairs of every range dealt round-robin to the devices, ONE
    in-order commit: the output equals `wtzmo -t 1` (the plain golden, not the -P stripes) for any number of devices."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--gpu-list", devs, "--batch", "16"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name,devs", [("zmo", "0,0"), ("dmo", "0,0"), ("zmo", "0,0,0"), ("dmo", "0,0,0"), ("zmo_S1", "0,0"), ("zmo_I", "0,0")])
def test_index_sharded_by_read_id_equals_unsharded(name, devs, emul_exe, tmp_path):
    """--shard-index (BASELINE configs[4]): every device indexes one contiguous read-id range; the frequency filter and the
    automatic cutoff use the k-mer counts of ALL shards (one exchange), every shard answers every query with its (read, strand) groups
    and the heap replay runs over the shards' lists in shard order.  The output must be the UNSHARDED golden (`zmo`, not `zmo_G2`:
    the reference's own -G filters each part by its own counts and gives different records)."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--gpu-list", devs, "--shard-index", "--batch", "16"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name", ["zmo", "dmo", "zmo_I", "zmo_G2"])
def test_per_batch_zindex_equals_all_reads_index(name, emul_exe, tmp_path):
    """--zindex-batch 1 (wtz_zindex_build_subset per batch: the batch's queries + every read in their candidate rows)"""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--zindex-batch", "1", "--batch", "7"])
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


@pytest.mark.parametrize("name,devs,extra", [("zmo", "0,0", []), ("dmo", "0,0,0", []), ("zmo_I", "0,0", []), ("zmo", "0,0,0", ["--shard-index"]), ("dmo", "0,0", ["--shard-index"]), ("zmo_n", "0,0", [])])
def test_per_batch_zindex_with_several_devices(name, devs, extra, emul_exe, tmp_path):
    """--zindex-batch 1 with --gpu-list (round 4: the two used to exclude each other): pairs are dealt by candidate id, so device d rebuilds, per
    batch, the candidate side of the batch's candidate reads = d (mod N) only, plus the query-side index of the batch's queries
    (wtz_zindex_build_queries); also with the k-mer index sharded (the BASELINE configs[4] combination)."""
    case = manifest()["cases"][name]
    md5, cont, _ = run_wtzmo_like(emul_exe, case, tmp_path, extra=["--gpu-list", devs, "--zindex-batch", "1", "--batch", "7"] + extra)
    assert md5 == case["md5_full"] and cont == case["md5_contained"]


def test_split_zindex_can_be_switched_off(emul_exe, tmp_path):
    """WTZ_NO_ZSPLIT=1: every device builds the z-mer index of all reads (the form before round 4); same records"""
    case = manifest()["cases"]["zmo"]
    env = dict(os.environ, WTZ_NO_ZSPLIT="1")
    out = os.path.join(str(tmp_path), "o.ovl")
    r = subprocess.run([emul_exe, "-i", os.path.join(ROOT, "tests", "golden", case["input"]), "-fo", out, "--gpu-list", "0,0", "--batch", "16"] + case["argv"], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    import hashlib
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["md5_full"]


@pytest.mark.parametrize("name", ["zmo", "dmo", "zmo_N"])
def test_writer_threads_keep_the_record_order(name, emul_exe, tmp_path):
    """Records are formatted by several writer threads and a regular file is written with positioned writes (pwritev at the chunk's offset, chunks
    side by side); a pipe is written strictly in turn.  With three records per chunk (WTZ_OUT_CHUNK_RECS, a test hook) hundreds of chunks are in
    flight: file, pipe and the sequential fallback must all be the reference's bytes."""
    import hashlib
    case = manifest()["cases"][name]
    env = dict(os.environ, WTZ_OUT_CHUNK_RECS="3")
    argv = [emul_exe, "-i", os.path.join(ROOT, "tests", "golden", case["input"]), "--batch", "16"] + case_argv(case, tmp_path)
    out = os.path.join(str(tmp_path), "f.ovl")
    subprocess.run(argv + ["-fo", out], check=True, capture_output=True, env=env)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["md5_full"]
    r = subprocess.run(argv + ["-fo", "-"], check=True, capture_output=True, env=env)          # a pipe
    assert hashlib.md5(r.stdout).hexdigest() == case["md5_full"]
    with open(os.path.join(str(tmp_path), "redir.ovl"), "wb") as fh:                           # stdout redirected to a regular file: positioned writes on fd 1
        subprocess.run(argv + ["-fo", "-"], check=True, stdout=fh, stderr=subprocess.DEVNULL, env=env)
    assert hashlib.md5(open(os.path.join(str(tmp_path), "redir.ovl"), "rb").read()).hexdigest() == case["md5_full"]
    subprocess.run(argv + ["-fo", out], check=True, capture_output=True, env=dict(env, WTZ_OUT_SEQUENTIAL="1"))
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["md5_full"]


def test_short_cigar_records_survive_a_lagging_writer(emul_exe, oracle_exe, tmp_path):
    """Accurate reads: most CIGAR texts are shorter than 256 bytes, the size below which the writer thread copies the text into its formatted run instead of
    pointing at it.  The copy happens on the writer thread, LATER than the commit - so a chunk of short records must hold the part's text buffer like a chunk of
    long ones does (round-4 regression: it did not, and a writer that lagged behind a slow consumer formatted text that the part had already refilled).
    WTZ_OUT_LAG_MS delays every chunk on the writer threads; two records per chunk and batches of four queries put the refills in front of the writer."""
    import hashlib
    from smartdenovo_amd import synth
    names, seqs = synth.synth_reads(40000, 18.0, seed=77, mean_len=2500.0, sigma=0.3, min_len=1200, err=0.002)
    fa = os.path.join(str(tmp_path), "hifi.fa"); synth.write_fasta(fa, names, seqs)
    argv = ["-i", fa, "-k", "16", "-s", "200", "-m", "0.6"]
    ref = os.path.join(str(tmp_path), "ora.ovl")
    subprocess.run([oracle_exe] + argv + ["-fo", ref], check=True, capture_output=True)
    want = open(ref, "rb").read()
    short = sum(1 for l in want.splitlines() if len(l.split(b"\t")[16]) < 256)
    assert want.count(b"\n") > 200 and short > 0.5 * want.count(b"\n")
    env = dict(os.environ, WTZ_OUT_CHUNK_RECS="2", WTZ_OUT_LAG_MS="15")
    for extra in ([], ["--gpus", "3"]):
        out = os.path.join(str(tmp_path), "lag.ovl")
        subprocess.run([emul_exe] + argv + ["--batch", "4", "-fo", out] + extra, check=True, capture_output=True, env=env)
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == hashlib.md5(want).hexdigest(), "records differ under a lagging writer (%s)" % " ".join(extra)
    r = subprocess.run([emul_exe] + argv + ["--batch", "4", "-fo", "-"], check=True, capture_output=True, env=env)      # a pipe: written strictly in turn
    assert r.stdout == want


def test_word_level_base_packing(tmp_path):
    """wtz_pack32 (two 64-bit loads + funnel shift per 32 bases, both strands, complement) == 32 single-base extractions."""
    exe = os.path.join(str(tmp_path), "check_pack32")
    subprocess.run(["g++", "-std=c++17", "-O1", "-DWTZ_EMUL", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "smartdenovo_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "emul", "check_pack32.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 bad" in r.stdout, r.stdout + r.stderr


REF_EXE = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")


@pytest.mark.skipif(not os.path.exists(REF_EXE), reason="reference binary not built (needs /root/reference once)")
@pytest.mark.parametrize("ndistinct,unrelated", [(748, True), (749, True), (748, False), (300, False)])
def test_pair_file_replay_with_repeated_preload_lines(ndistinct, unrelated, emul_exe, tmp_path):
    """-9 lists the tested pairs in the iteration order of the reference's hash set, and a REPEATED -L line is still a put_u64hash there: its
    capacity check runs before the lookup (hashset.h:224-226, 351), so a duplicate arriving when the table is exactly full (748 keys in the
    initial 1117 slots) grows it although no key is added.  `unrelated`: 40 random reads that share nothing, so the run itself tests no new pair and
    the table the file is written from is the one the preload left (the case that tells the two behaviours apart); otherwise the tiny read
    set, where new pairs follow the preload.  Live comparison with the compiled reference."""
    import gzip
    import random
    if unrelated:
        rng = random.Random(7)
        fa = os.path.join(str(tmp_path), "unrelated.fa")
        names = ["u%02d" % i for i in range(40)]
        with open(fa, "w") as fh:
            for n in names:
                fh.write(">%s\n%s\n" % (n, "".join(rng.choice("ACGT") for _ in range(3000 + rng.randrange(500)))))
    else:
        fa = os.path.join(ROOT, "tests", "golden", "tiny.fa.gz")
        names = [ln[1:].split()[0] for ln in gzip.open(fa, "rt") if ln.startswith(">")]
    pairs = [(names[i], names[j]) for i in range(len(names)) for j in range(i + 1, len(names))][:ndistinct]
    assert len(pairs) == ndistinct
    lst = os.path.join(str(tmp_path), "tested.txt")
    with open(lst, "w") as fh:
        for a, b in pairs + pairs[:5]:
            fh.write("%s\t%s\n" % (a, b))
    argv = ["-i", fa, "-k", "16", "-s", "200", "-m", "0.6", "-L", lst]
    outs = []
    for tag, exe, extra in (("ref", REF_EXE, ["-t", "1"]), ("emul", emul_exe, ["--batch", "16"])):
        o, p9 = os.path.join(str(tmp_path), tag + ".ovl"), os.path.join(str(tmp_path), tag + ".pairs")
        subprocess.run([exe] + extra + argv + ["-fo", o, "-9", p9], check=True, capture_output=True)
        outs.append((open(o, "rb").read(), open(p9, "rb").read()))
    assert outs[0][0] == outs[1][0], ".ovl differs from the reference with -L"
    assert outs[0][1] == outs[1][1], "-9 pair file is not in the reference's hash-set order"
    assert outs[0][1].count(b"\n") >= ndistinct
