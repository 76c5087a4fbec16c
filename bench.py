#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: Gbp of candidate pairs aligned per second (wtzmo all-vs-all).

One "step" = one complete overlap phase of the drop-in wtzmo (z-mer index + k-mer index build, candidate search, pair seeding /
windows / chaining, banded SW, in-order commit, .ovl writing) over the synthetic read set of BASELINE.json configs[2] - the
configuration the metric is quoted on: 1.2 Gbp of PacBio-shape reads (12 Mbp iid genome x100, lognormal mean 10 kb, 15 % error,
seed 29) with the 2-bit reads and both indexes RESIDENT IN HBM of one MI355X (`--workload ecoli` = configs[1], 115 Mbp).  The output of
exactly this input is pinned to the reference: tests/test_gpu_scale.py compares its md5 with `wtzmo -t 1` (tests/golden/big_manifest.json).
The host driver runs in-process (libwtzmo_host.so = the C `wtzmo` main built as a shared object) so that exactly K steps are
bracketed by barrier + torch.cuda.synchronize() on both sides.

N > 1: one process per GPU - launched by the driver under torch.distributed.run, or by `python bench.py --gpus N` itself when no launcher set
WORLD_SIZE (it re-executes itself under torch.distributed.run; --gpus must equal the number of ranks).  The order-dependent part of `wtzmo -t 1`
is one sequential stream by definition, so rank 0 plans and commits; the pure device stages (seed lookup per query; pair seeding, windows,
banded SW and CIGAR rendering per pair) are dealt over the ranks - pairs by candidate id, so that a rank's z-mer index holds the candidate side
of its own residue class of the reads plus the queries of the batch in flight; reads and k-mer index replicated in every HBM (or the k-mer
index sharded by read id: --workload human30); requests go out from rank 0 and results
(pair summaries, window boxes, alignment results, CIGAR text) come back with RCCL send / recv over xGMI inside the timed region
(smartdenovo_amd/multigpu.py).  Rank 0 writes ONE .ovl, identical to `wtzmo -t 1` for any N; total work is fixed -> "strong" scaling.

numerator  = sum over the timed steps of (len(a)+len(b)) over pairs that entered pair alignment (the same number for every N)
value      = numerator / wall seconds of the K timed steps (max over ranks) / 1e9
roofline      = K-sw3 (shifting-band extension, the dominant DP stage; two concurrent kernels): DP cells exactly as the reference loops
                execute them x 12 int32 ops per cell / HIP-event time of the stage launches, against the int32 VALU peak
roofline_sw1  = K-sw1 (fixed-band extension between anchors, K_winalign), roofline_sw2 = K-sw2 (global banded, K_gap): same pricing
roofline_seed = seed lookup (K_candidates): algorithmic bytes (L/4 + 16 B per probe + 4 B per seed entry) / kernel time vs 8 TB/s
cpu_baseline  = the REAL reference `wtzmo -t 32` (oracle/_ref, prebuilt) on THE BENCH INPUT ITSELF for configs[1] / configs[2] (a same-shape sample for the larger
                shapes), its overlap phase timed like the GPU's (from "calculating overlaps" to exit; FASTA load excluded); cpu_baseline_all_cores = the same
                binary with -t <all hardware threads> on a bounded sample (north_star's figure: slower than -t 32, the reference's index build is O(threads x bases))
parity        = md5 of the last step's .ovl against reference `wtzmo -t 1` on the same input (tests/golden/big_manifest.json); a mismatch voids the run (exit 1)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ZMO = ["-k", "16", "-s", "200", "-m", "0.6"]
DMO = ["-k", "16", "-z", "10", "-Z", "16", "-U", "-1", "-m", "0.1", "-A", "1000"]
INT32_VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tint32op/s
HBM_PEAK_GBS = 8000.0
OPS_PER_CELL = 12                                        # SURVEY 8d nominal op count of one DP cell update
# seeds / sizes are those of tests/golden/big_manifest.json, so the md5 parity of exactly these inputs is a driver-run test
WORKLOADS = {
    "yeast100": dict(genome=12000000, coverage=100.0, seed=29, cpu_genome=12000000, golden="yeast100",      # the CPU baseline runs on the bench input itself (~200 s at -t 32)
                     name="BASELINE configs[2]: 1.2 Gbp of synthetic PacBio-shape reads (12 Mbp iid genome x100)"),
    "ecoli": dict(genome=4600000, coverage=25.0, seed=11, cpu_genome=4600000, golden="ecoli",      # the CPU baseline runs on the bench input itself (about 14 s at -t 32)
                  name="BASELINE configs[1]: E. coli-shape synthetic PacBio reads (4.6 Mbp iid genome x25)"),
    # configs[3] shape: 951 827 reads / 9.8 Gbp (generating the 10 GB FASTA takes ~10 minutes of numpy before the first step; the z-mer index is rebuilt per batch of
    # queries above 2.4 Gbp per device).  No whole-job reference exists (`wtzmo -t 1` would run for days): parity at this shape is pinned on the reference's
    # -P 128 -p 0 stripe (tests/test_gpu_scale.py[fly70_zmo_P128p0]); the CPU baseline is a same-shape sample.
    "fly70": dict(genome=140000000, coverage=70.0, seed=53, cpu_genome=1400000, golden=None, extra=[], stripe="fly70_zmo_P128p0",
                  name="BASELINE configs[3]: D. melanogaster-shape synthetic reads (140 Mbp iid genome x70 = 9.8 Gbp), query-sharded"),
    # configs[4] shape, scaled to what one box can generate in minutes: 30x of an iid genome with the human pipeline's -k 17 (smartdenovo.pl:16), the k-mer index
    # sharded by read-id range over the ranks (--shard-index) and the z-mer index per batch; --genome 3000000000 is the full configs[4] (90 Gbp: needs 8 GPUs)
    "human30": dict(genome=100000000, coverage=30.0, seed=59, cpu_genome=2000000, golden=None, extra=["--shard-index", "--zindex-batch", "1"], k17=True, stripe="human30_zmo_P64p0",
                    name="BASELINE configs[4] shape, scaled: 30x of a 100 Mbp iid genome (3 Gbp of reads), -k 17, per-GPU index shard"),
}


def file_md5(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def kernel_source_id():
    """sha1 over the CODE of the gfx950 code objects the library was built with: the .text (ISA), .rodata (kernel descriptors) and .note (kernel metadata: registers,
    LDS, scratch) sections of every amdgcn entry of the clang offload bundle in the .hip_fatbin section of smartdenovo_amd/libwtzmo_hip.so.  That is what a PMC summary
    under profiles/ was measured on and what this run executes.  Round 4 hashed the source files whole (a host-only edit of wtz_lib.cpp made the driver's line call
    its evidence stale although no kernel had changed); the first round-5 form hashed the whole .hip_fatbin - and two builds of the SAME sources differ there: the
    order of the code object's dynamic symbol table is not deterministic (.dynsym / .dynstr / .strtab differ, .text / .rodata / .note do not: checked on four builds,
    one of them in another directory).  A rebuild of unchanged sources - the driver's at round end - now gives the same id."""
    import hashlib
    import struct

    def sections(elf):
        shoff, = struct.unpack_from("<Q", elf, 0x28); shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
        sec = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
        stro = sec[shstrndx][4]
        return {elf[stro + name:elf.index(b"\0", stro + name)]: (off, size, typ) for name, typ, _f, _a, off, size, *_ in sec}

    lib = os.path.join(ROOT, "smartdenovo_amd", "libwtzmo_hip.so")
    try:
        b = open(lib, "rb").read()
        off, size, _t = sections(b)[b".hip_fatbin"]
        fb = b[off:off + size]
        if fb[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
            return "unbuilt"
        n, = struct.unpack_from("<Q", fb, 24); p = 32; h = hashlib.sha1(); found = False
        for _ in range(n):
            eoff, esize, tl = struct.unpack_from("<QQQ", fb, p); p += 24; triple = fb[p:p + tl]; p += tl
            if esize and b"amdgcn" in triple:
                elf = fb[eoff:eoff + esize]; sc = sections(elf)
                for nm in (b".text", b".rodata", b".note"):
                    if nm in sc and sc[nm][2] != 8:
                        o, z, _ = sc[nm]; h.update(nm); h.update(elf[o:o + z]); found = True
        return h.hexdigest()[:16] if found else "unbuilt"
    except (OSError, struct.error, ValueError, KeyError):
        pass
    return "unbuilt"


def git_blob_id(path):
    import hashlib
    b = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(b) + b).hexdigest()


def gen_reads(path, genome, coverage, seed):
    from smartdenovo_amd import synth
    if os.path.exists(path) and os.path.exists(path + ".meta"):
        return json.load(open(path + ".meta"))
    names, seqs = synth.synth_reads(genome, coverage, seed=seed)
    md5 = synth.write_fasta(path + ".tmp", names, seqs)
    os.replace(path + ".tmp", path)
    meta = {"reads": len(names), "bases": int(sum(s.size for s in seqs)), "md5": md5}
    json.dump(meta, open(path + ".meta", "w"))
    return meta


def cpu_baseline(engine_argv, genome, coverage, seed, tmp, same_input=False, threads=0, fa=None):
    """Reference (or oracle port) timed on this host's cores on a bounded sample of the same workload shape."""
    ref = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")
    ora = os.path.join(ROOT, "oracle", "wtzmo_oracle")
    fa = fa or os.path.join(tmp, "cpu_sample_G%d_c%g_s%d.fa" % (genome, coverage, seed))
    meta = gen_reads(fa, genome, coverage, seed)
    lens = {}
    name = None
    for line in open(fa):
        if line[0] == ">":
            name = line[1:].strip()
        else:
            lens[name] = len(line.strip())
    # the reference's index build is O(threads x bases) (every thread scans every read, wtzmo.c:272): on the 256-thread GPU host
    # -t 256 is 4x SLOWER than -t 32 (56 s vs 13.8 s on the full E. coli-shape set), so the baseline uses min(32, cores)
    ncpu = threads or min(32, os.cpu_count() or 1)
    out = os.path.join(tmp, "cpu.ovl")
    if os.path.exists(ref):
        pairs = os.path.join(tmp, "cpu.pairs")
        t0 = time.perf_counter()
        pr = subprocess.Popen([ref, "-t", str(ncpu), "-i", fa, "-fo", out, "-9", pairs] + engine_argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        t_ovl = None
        for line in pr.stderr:          # "[date] calculating overlaps, N threads" opens overlap_wtzmo (index build included, FASTA load excluded)
            if t_ovl is None and b"calculating overlaps" in line:
                t_ovl = time.perf_counter()
        rc = pr.wait()
        t1 = time.perf_counter()
        if rc != 0 or t_ovl is None:
            raise RuntimeError("reference wtzmo failed (rc %d)" % rc)
        dt = t1 - t_ovl
        bp = 0
        for line in open(pairs):
            a, b = line.split()
            bp += lens[a] + lens[b]
        return {"value": bp / dt / 1e9, "unit": "Gbp pair-bp/s", "cores": ncpu, "kind": "reference",
                "sample": "reference wtzmo -t %d on %s, %d bp genome x%g (%d reads, %d bp): overlap phase %.2f s (whole process %.2f s incl. FASTA load), %d pair-bp"
                          % (ncpu, "THE BENCH INPUT ITSELF" if same_input else "a bounded sample of the same generator (same coverage, another seed)", genome, coverage, meta["reads"], meta["bases"], dt, t1 - t0, bp)}
    if not os.path.exists(ora):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "wtzmo_oracle"], check=True)
    st = os.path.join(tmp, "cpu.stats")
    subprocess.run([ora, "-i", fa, "-fo", out, "--stats", st] + engine_argv, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n, bp, secs = open(st).read().split()[:3]
    return {"value": int(bp) / float(secs) / 1e9, "unit": "Gbp pair-bp/s", "cores": 1, "kind": "port",
            "sample": "oracle port (1 thread, -t 1 semantics) on synthetic %d bp genome x%g (%d reads), overlap phase %.2f s" % (genome, coverage, meta["reads"], float(secs))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="yeast100", help="yeast100 = BASELINE configs[2] (1.2 Gbp of reads), ecoli = configs[1]")
    ap.add_argument("--genome", type=int, default=0, help="override: synthetic genome length")
    ap.add_argument("--coverage", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--engine", choices=["zmo", "dmo"], default="zmo")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--pool-gb", type=int, default=0, help="device scratch (0 = the library's default: 45 %% of the free HBM, at most 128 GB)")
    ap.add_argument("--cpu-genome", type=int, default=0, help="genome length of the bounded CPU-baseline sample (same coverage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["input", "sample"], default=None, help="reference wtzmo on the bench input itself (default where the workload has a golden) or on a bounded same-shape sample")
    ap.add_argument("--no-verify", action="store_true", help="skip the md5 comparison of the last step's .ovl with the reference golden (tests/golden/big_manifest.json)")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    a.genome = a.genome or wl["genome"]; a.coverage = a.coverage or wl["coverage"]; a.seed = a.seed or wl["seed"]
    same_as_golden = (a.genome == wl["genome"] and a.coverage == wl["coverage"] and a.seed == wl["seed"])
    if a.cpu_baseline is None:
        a.cpu_baseline = "input" if (wl.get("golden") and same_as_golden) else "sample"
    a.cpu_genome = a.cpu_genome or (a.genome if a.cpu_baseline == "input" else min(wl["cpu_genome"], a.genome) if wl["cpu_genome"] != wl["genome"] else max(200000, a.genome // 20))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher - one rank per GPU under torch.distributed.run (the driver's own
        # multi-GPU command form), rank 0 prints the one JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (a.gpus, " ".join(cmd)), file=sys.stderr)
        sys.exit(subprocess.run(cmd).returncode)
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: the two must agree (the JSON line reports n_gpus = the number of ranks that ran)" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the hot path has no CPU fallback")
    # WTZ_BENCH_BACKEND=gloo: a test aid for boxes with fewer GPUs than ranks (RCCL refuses two ranks on one device): ranks share
    # the GPUs round-robin and exchange through gloo / host tensors; everything else is the measured path
    backend = os.environ.get("WTZ_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks on %d visible GPU(s): RCCL needs one device per rank (WTZ_BENCH_BACKEND=gloo lets ranks share a device - a test aid, not a measurement)" % (world, torch.cuda.device_count()))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        long = datetime.timedelta(hours=3)      # rank 0 may generate a multi-GB synthetic read set before the first barrier
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=long)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=long)

    import __graft_entry__ as ge
    from smartdenovo_amd import multigpu
    if rank == 0 and not (os.path.exists(ge.HOSTLIB) and os.path.exists(ge.LIB)):
        ge.build_product()
    tmp = os.environ.get("WTZ_BENCH_TMP", os.path.join(tempfile.gettempdir(), "wtz_bench"))
    os.makedirs(tmp, exist_ok=True)
    fa = os.path.join(tmp, "reads_G%d_c%g_s%d.fa" % (a.genome, a.coverage, a.seed))
    meta = None
    if rank == 0:
        meta = gen_reads(fa, a.genome, a.coverage, a.seed)
    if dist:
        dist.barrier()
    if meta is None:
        meta = json.load(open(fa + ".meta"))

    eng = list(ZMO if a.engine == "zmo" else DMO)
    if wl.get("k17"):
        eng[eng.index("-k") + 1] = "17"
    drv_extra = [x for x in wl.get("extra", []) if not (x == "--shard-index" and world == 1)]      # a sharded index needs more than one rank
    if rank == 0:
        # --repeat keeps at most ONE stale output next to the file being written (wtzmo_main.c: stale_job), whatever W + K is; an .ovl with
        # CIGARs is ~3 bytes per read base (3.4 GB at configs[2]).  r02's driver run died of ENOSPC after 20 kept files: check, and say so.
        need = 2 * 3 * meta["bases"] + (1 << 30)
        free = shutil.disk_usage(tmp).free
        print("[bench] %s: %.1f GB free, this run needs <= %.1f GB (two outputs of ~%.1f GB at any time, independent of --steps)"
              % (tmp, free / 1e9, need / 1e9, 3 * meta["bases"] / 1e9), file=sys.stderr)
        if free < need:
            sys.exit("bench.py: not enough free space under %s (%.1f GB free, %.1f GB needed); set WTZ_BENCH_TMP" % (tmp, free / 1e9, need / 1e9))
    out = os.path.join(tmp, "bench_r%d.ovl" % rank)
    for stale in (out, out + ".prev"):          # leftovers of a run that died
        if os.path.exists(stale):
            os.remove(stale)
    stats = os.path.join(tmp, "bench_r%d.stats" % rank)
    W, K = a.warmup, a.steps
    dev_args = ["--gpu", str(local)]
    if os.environ.get("WTZ_BENCH_CONTEXTS"):      # experiment: several contexts on this rank's device (--gpu-list d,d: every range dealt over them, their launches side by side)
        dev_args = ["--gpu-list", ",".join([str(local)] * int(os.environ["WTZ_BENCH_CONTEXTS"]))]
    argv = ["wtzmo"] + dev_args + ["-i", fa, "-fo", out, "--repeat", str(W + K), "--stats", stats] + (["--pool-gb", str(a.pool_gb)] if a.pool_gb else []) + drv_extra + eng
    if a.max_batch:
        argv += ["--batch", str(a.max_batch)]
    if os.environ.get("WTZ_BENCH_FIRST_BATCH"):      # experiment: size of the first batch of the ramp (256 -> x4 -> ... -> --batch)
        argv += ["--first-batch", os.environ["WTZ_BENCH_FIRST_BATCH"]]
    host = C.CDLL(ge.HOSTLIB)
    xchg = None
    if dist:
        xchg = multigpu.RankExchange(dist, "cuda" if backend == "nccl" else "cpu")
        xchg.install(host)
    T = {"t0": None, "t1": None, "gathered": 0}
    HOOK = C.CFUNCTYPE(None, C.c_int, C.c_int)

    def hook(rep, phase):
        if phase == 0 and rep == W:
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            T["t0"] = time.perf_counter()
        if phase == 1:
            if rep == W + K - 1:
                torch.cuda.synchronize()
                if dist:
                    dist.barrier()
                T["t1"] = time.perf_counter()

    cb = HOOK(hook)
    host.wtzmo_set_hook(cb)
    cargv = (C.c_char_p * (len(argv) + 1))(*[s.encode() for s in argv], None)
    host.wtzmo_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    rc = host.wtzmo_main(len(argv), cargv)
    if rc != 0:
        sys.exit("wtzmo_main returned %d" % rc)
    dt = T["t1"] - T["t0"]
    rows = [l.split("\t") for l in open(stats).read().strip().split("\n")]
    timed = rows[W:W + K]
    pair_bp = sum(int(r[1]) for r in timed)
    n_pairs = sum(int(r[0]) for r in timed)
    last = timed[-1]
    if dist:
        xdev = "cuda" if backend == "nccl" else "cpu"
        v = torch.tensor([float(pair_bp), float(n_pairs)], dtype=torch.float64, device=xdev)
        dist.all_reduce(v)
        pair_bp, n_pairs = int(v[0].item()), int(v[1].item())
        d = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        dt = float(d.item())
    sent_dev = 0
    if dist:      # what the peers sent straight from device memory (the CIGAR text: multigpu.py::_send_dev), summed over the ranks: rank 0 only receives
        sd = torch.tensor([float(xchg.bytes_sent_from_device)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(sd)
        sent_dev = int(sd.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    value = pair_bp / dt / 1e9
    ms = {k: float(last[4 + i]) for i, k in enumerate(["index", "zindex", "candidates", "pairs", "winalign", "stitch"])}
    cells_shift, cells_fixed, cells_global = int(last[10]), int(last[11]), int(last[12])
    seed_bytes = int(last[13])
    ms_ext = float(last[15])
    ms_gap = float(last[19]) if len(last) > 19 else 0.0
    n_ranges, n_split = (int(last[20]), int(last[21])) if len(last) > 21 else (0, 0)
    zmer_bytes = int(last[22]) if len(last) > 22 else 0
    ing_ms, ing_bytes = (float(last[23]), int(last[24])) if len(last) > 24 else (0.0, 0)
    host_s = None
    if len(last) > 32:      # host seconds of the last step: what tools/design_numbers.py builds the N = 8 budget of DESIGN section 7 from
        host_s = {"wall": float(last[2]), "index_builds": float(last[3]), "device_stage_calls": float(last[25]), "commit": float(last[26]),
                  "commit_sections": {"rows_closed_filter_order": float(last[27]), "window_depth_seed_weights": float(last[28]), "hits": float(last[29]), "plan_pairs": float(last[30])},
                  "batches": int(last[31]), "ranges": n_ranges, "queries_planned": int(last[32]), "queries_used": int(last[17])}

    def valu_roofline(kernel, cells, ms_k, extra=None):
        ach = cells * OPS_PER_CELL / (ms_k * 1e-3) / 1e12 if ms_k > 0 else None
        r = {"kernel": kernel, "bound": "valu_int32", "achieved": ach, "peak": INT32_VALU_PEAK_TOPS, "unit": "Tint32op/s",
             "frac": ach / INT32_VALU_PEAK_TOPS if ach is not None else None, "cell_updates_per_s": cells / (ms_k * 1e-3) if ms_k > 0 else None,
             "cells_per_step": cells, "kernel_ms_per_step": ms_k, "ops_per_cell": OPS_PER_CELL, "traffic": None}
        r.update(extra or {})
        return r

    res = {
        "metric": "Gbp of candidate pairs aligned/sec (wtzmo all-vs-all)",
        "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32/int16", "data": "synthetic",
        "config": {"workload": "%s: %d bp iid genome x%g, seed %d, lognormal mean 10 kb, 15%% error (ins:del:sub 50:30:20)" % (WORKLOADS[a.workload]["name"], a.genome, a.coverage, a.seed),
                   "reads": meta["reads"], "read_bases": meta["bases"], "engine": a.engine, "argv": " ".join(eng),
                   "parallelism": "1 GPU" if world == 1 else "%d ranks (%s): pairs dealt by candidate id, seed-lookup requests round-robin (reads + k-mer index replicated%s; z-mer index: candidate side of the rank's "
                                                              "residue class + the batch's queries), central in-order commit on rank 0, results gathered with send/recv" % (world, "RCCL" if backend == "nccl" else backend + " - a test aid", ", k-mer index sharded by read id" if "--shard-index" in drv_extra else ""),
                   "parity": "this input's .ovl md5 == reference `wtzmo -t 1` (tests/test_gpu_scale.py, tests/golden/big_manifest.json)%s" % ("" if world == 1 else "; any N writes the same file (tests/test_multi_rank_gloo.py)"),
                   "scratch": "%d batch ranges planned to the pool, %d split after a pool overflow" % (n_ranges, n_split)},
        "pairs_per_step": n_pairs // K, "pair_bp_per_step": pair_bp // K, "records_last_step": int(last[14]),
        "kernel_ms_last_step": dict(ms, ksw3_wave=ms_ext, ksw2_gap=ms_gap),
        "host_seconds_last_step": host_s,
        "roofline": valu_roofline("K-sw3 shifting-band extension (kswx_extend_align_shift_core), DP rows in registers in the anti-diagonal frame: wtz_kernel_stitch_ext_pk - both end extensions of a stitched "
                                  "overlap and the join between them on one wavefront (wtz_stitch_fused.h), TWO 16-bit cells per vector register (wtz_sw_frame16.h, round 6) -, beside it wtz_kernel_stitch_ext_fr (the 32-bit "
                                  "frame form, wtz_sw_frame.h) for the items outside the 16-bit window, wtz_kernel_extjobs for what is outside every register form's envelope; time = HIP events around the launches of the stage; "
                                  "the roof stays the int32 one the earlier rounds were priced against (12 ops per cell at 78.6 Tint32op/s) - against the packed-int16 rate of the VALU (twice that) the fraction is half",
                                  cells_shift, ms_ext, {"backtrack_GBps": cells_shift / (ms_ext * 1e-3) / 1e9 if ms_ext > 0 else None, "arithmetic": "int16 pairs (v_pk_*), int32 where a job's values leave a 16-bit window"}),
        "roofline_sw1": valu_roofline("K-sw1 fixed-band extension between anchors (kswx_extend_align_core), one LANE per problem: K_lplan -> K_ldp (register DP, relative mode) -> "
                                      "K_ltb (traceback) -> K_lfold (score chain, z-mer runs, CIGAR) + the chained wave kernel for the windows the fold leaves; the time is the whole stage's", cells_fixed, ms["winalign"]),
        "roofline_sw2": valu_roofline("K-sw2 global banded alignment of the gaps between windows (ksw_global2 incl. every band-doubling call): one lane per gap (K_gplan / K_gdp / K_gtb), "
                                      "the rest in wtz_kernel_coop_tasks<K_gap> (+ K_gap_wide)", cells_global, ms_gap),
        "roofline_zmer": {"kernel": "wtz_kernel_coop_tasks<K_pair> (z-mer matching hzm_aln.h:173-224 + windows / chain or dot-matrix per pair; the time is the whole kernel's)", "bound": "hbm",
                          "achieved": zmer_bytes / (ms["pairs"] * 1e-3) / 1e9 if ms["pairs"] > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (zmer_bytes / (ms["pairs"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms["pairs"] > 0 else None,
                          "algorithmic_bytes_per_step": zmer_bytes, "kernel_ms_per_step": ms["pairs"], "traffic": None},
        "roofline_seed": {"kernel": "wtz_kernel_wg_tasks<K_candidates_wg> (hzm seed lookup: one workgroup per query, tuples partitioned by target read, per-bucket LDS sort, candidate heap)", "bound": "hbm",
                          "achieved": seed_bytes / (ms["candidates"] * 1e-3) / 1e9 if ms["candidates"] > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (seed_bytes / (ms["candidates"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms["candidates"] > 0 else None,
                          "algorithmic_bytes_per_step": seed_bytes, "kernel_ms_per_step": ms["candidates"], "traffic": None},
    }
    if ing_ms > 0:      # f4: FASTA text -> 2-bit BaseBank on the device, ONCE at load time (outside the timed steps: the metric excludes the FASTA load)
        res["roofline_ingest"] = {"kernel": "wtz_kernel_pack_ascii (+ K_pack_fix for non-ACGT bytes): seq2basebank dna.h:397-410 on the device, measured once at load time, not inside the timed steps",
                                  "bound": "hbm", "achieved": ing_bytes / (ing_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ing_bytes / (ing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "algorithmic_bytes": ing_bytes, "kernel_ms": ing_ms, "traffic": None}
    # the nominal peak above is SURVEY 8d's formula (CUs x lanes x clock); what gfx950 ISSUES was measured with tools/ubench/valu_int32.hip
    # (profiles/r03_valu_int32_ubench.txt): v_add / v_sub ~60 Tlane-op/s, v_max_i32 / v_alignbit / v_mad_i32_i24 / v_lshl_or ~36.6 (half rate)
    for key in ("roofline", "roofline_sw1", "roofline_sw2"):
        res[key]["peak_measured_Tops"] = {"v_add_u32": 63.5, "v_max_i32": 36.6}
    # HBM traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command with --steps 1 --warmup 0 (so totals are per step),
    # condensed by tools/summarize_profiles.py into profiles/ (FETCH_SIZE doubled as the gfx950 guide says).  `traffic` is per LAUNCH like
    # `achieved` (total / dispatches), `traffic_per_step` the total next to the per-step algorithmic bytes.  A PMC pass cannot run inside a timed
    # bench run, so the numbers come from the newest committed summary of this workload / engine; its sidecar (<csv>.meta.json, written by
    # tools/gpu_r04_final.sh) names the kernel sources it was measured on, and the line says whether they are the sources of THIS build.
    import csv
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_%s_pmc_per_kernel.csv" % (a.workload, a.engine))))
    src = cands[-1] if cands else None
    ksrc_now = kernel_source_id()
    res["build"] = {"kernel_source_id": ksrc_now, "lib_md5": file_md5(ge.LIB)}
    if src is None:
        res["traffic_source"] = None
    else:
        prov = {"file": os.path.relpath(src, ROOT), "git_blob": git_blob_id(src), "measured_on_kernel_source_id": None, "same_kernel_sources_as_this_build": None}
        if os.path.exists(src + ".meta.json"):
            prov["measured_on_kernel_source_id"] = json.load(open(src + ".meta.json")).get("kernel_source_id")
            prov["same_kernel_sources_as_this_build"] = (prov["measured_on_kernel_source_id"] == ksrc_now)
        res["traffic_source"] = prov
        kmap = {"wtz_kernel_stitch_ext_pk": "roofline", "wtz_kernel_extjobs_pk": "roofline", "wtz_kernel_stitch_ext_fr": "roofline", "wtz_kernel_extjobs_fr": "roofline", "wtz_kernel_extjobs_reg": "roofline", "wtz_kernel_extjobs_mw": "roofline", "wtz_kernel_extjobs": "roofline", "K_winalign": "roofline_sw1", "K_ldp": "roofline_sw1", "K_ltb": "roofline_sw1", "K_lfold": "roofline_sw1", "K_lplan": "roofline_sw1",
                "K_gap": "roofline_sw2", "K_gdp": "roofline_sw2", "K_gtb": "roofline_sw2", "K_candidates_wg": "roofline_seed", "K_candidates": "roofline_seed", "K_pair": "roofline_zmer", "K_pair_dm": "roofline_zmer", "K_pair_big": "roofline_zmer"}
        for row in csv.DictReader(open(src)):
            key = kmap.get(row["kernel"])
            if key is None:
                continue
            tot = float(row["hbm_bytes_est"])
            R = res[key]
            R["traffic_per_step"] = (R.get("traffic_per_step") or 0.0) + tot
            R["traffic"] = (R["traffic"] or 0.0) + tot / max(1, int(row["dispatches"]))
            R["traffic_note"] = ("PMC (2 x FETCH_SIZE + WRITE_SIZE) of a --steps 1 run of this command, read from %s (git blob %s; not measured in this run; measured on %s kernel sources)"
                                 % (prov["file"], prov["git_blob"][:12], "THIS build's" if prov["same_kernel_sources_as_this_build"] else ("OTHER (stale)" if prov["same_kernel_sources_as_this_build"] is False else "unrecorded")))
            if key not in ("roofline_seed", "roofline_zmer") and row.get("SQ_INSTS_VALU") and R["kernel_ms_per_step"] > 0:
                # wave-level VALU instructions of one step x 2 issue cycles on a SIMD-32 / SIMD-cycles of the live kernel time
                R["valu_issue_frac_pmc"] = (R.get("valu_issue_frac_pmc") or 0.0) + float(row["SQ_INSTS_VALU"]) * 2.0 / (R["kernel_ms_per_step"] * 1e-3 * 2.4e9 * 256 * 4)
        for key in ("roofline", "roofline_sw1", "roofline_sw2"):
            if res[key].get("traffic_per_step"):
                res[key]["algorithmic_trace_bytes_per_step"] = res[key]["cells_per_step"]       # the reference's layout: 1 trace byte per cell
    if world > 1:
        res["gathered_result_bytes_per_step"] = xchg.bytes_received // (W + K)
        res["exchange_messages_per_step"] = xchg.messages // (W + K)
        res["bytes_sent_from_device"] = sent_dev      # all ranks, all steps: > 0 iff the device-resident send path carried the text
        res["backend"] = backend
    # parity of THIS run: the last step's file against the md5 of reference `wtzmo -t 1` on the same input (generated once, committed)
    gold = None
    if wl.get("golden") and same_as_golden and not a.no_verify:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "big_manifest.json")))["cases"].get("%s_%s" % (wl["golden"], a.engine))
    if gold is not None:
        got = file_md5(out)
        res["parity"] = {"md5": got, "reference_md5": gold["md5_full"], "match": got == gold["md5_full"], "records": int(last[14]), "reference_records": gold["records"]}
        if not res["parity"]["match"]:
            print(json.dumps(res))
            sys.exit("bench.py: the .ovl of the last step differs from the reference golden (%s vs %s): the number above is void" % (got, gold["md5_full"]))
    elif wl.get("stripe") and same_as_golden and not a.no_verify and a.engine == "zmo":
        # no whole-job reference exists at this size (`wtzmo -t 1` would run for days): the reference's own query stripe (-P n -p 0: every n-th query against the
        # FULL index, wtzmo.c:1291,1314) is pinned in big_manifest.json; a second, short pass of the drop-in over the same resident input - same index form as the
        # timed steps (sharded over two contexts of this rank's device where the workload shards it) - must write its bytes.  Outside the timed region.
        sc = json.load(open(os.path.join(ROOT, "tests", "golden", "big_manifest.json")))["cases"][wl["stripe"]]
        out2 = os.path.join(tmp, "bench_stripe_r%d.ovl" % rank)
        wx = list(wl.get("extra", []))
        form = (["--gpu-list", "%d,%d" % (local, local), "--pool-gb", "48"] if "--shard-index" in wx else ["--gpu", str(local)])
        t0s = time.perf_counter()
        r = subprocess.run([ge.EXE] + form + ["-i", fa, "-fo", out2] + wx + sc["argv"], capture_output=True)
        dts = time.perf_counter() - t0s
        got = file_md5(out2) if r.returncode == 0 and os.path.exists(out2) else None
        res["parity"] = {"kind": "reference stripe %s" % " ".join(sc["argv"][-4:]), "md5": got, "reference_md5": sc["md5_full"], "match": got == sc["md5_full"],
                         "reference_records": sc["records"], "stripe_wall_s": round(dts, 1), "reference_seconds_t1": sc.get("reference_seconds"),
                         "note": "no whole-job reference output exists at this size; the stripe is the reference's own -P/-p job striping against the full index"}
        for f in (out2, out2 + ".contained"):
            if os.path.exists(f):
                os.remove(f)
        if not res["parity"]["match"]:
            print(json.dumps(res))
            sys.exit("bench.py: the reference stripe differs (%s vs %s; rc %d): %s" % (got, sc["md5_full"], r.returncode, r.stderr.decode()[-500:]))
    else:
        res["parity"] = {"match": None, "note": "no whole-job reference output exists for this input (parity at this shape: tests/test_gpu_scale.py)"}
    if world == 1 and not a.no_cpu_baseline:
        try:
            same = (a.cpu_genome == a.genome)
            res["cpu_baseline"] = cpu_baseline(eng, a.cpu_genome, a.coverage, a.seed if same else a.seed + 1000, tmp, same_input=same, fa=fa if same else None)
            res["cpu_baseline"]["threads_note"] = ("-t %d, not all %d hardware threads: the reference's index build is O(threads x bases) (every thread scans every read, wtzmo.c:272) - "
                                                   "measured on this host, E. coli shape: -t 256 is 4x slower than -t 32" % (res["cpu_baseline"]["cores"], os.cpu_count() or 1))
            if (os.cpu_count() or 1) > res["cpu_baseline"]["cores"]:
                # north_star's "-t <all host cores>", stated too: on a bounded sample of the same shape (the full input at -t 256 runs for a quarter of an hour)
                sg = a.cpu_genome if a.cpu_genome <= 1000000 else max(200000, a.genome // 20)
                allc = cpu_baseline(eng, sg, a.coverage, a.seed + 1000, tmp, same_input=False, threads=os.cpu_count())
                res["cpu_baseline_all_cores"] = allc
        except Exception as e:      # the baseline is reported, never required
            res["cpu_baseline"] = {"value": None, "error": str(e)}
    if a.engine == "dmo":       # no banded SW in the dot-matrix engine (SURVEY finding 2): its dominant kernel is K_pair, priced against HBM (after the PMC traffic was attached above)
        res["roofline"] = dict(res["roofline_zmer"])
    print(json.dumps(res))
    for leftover in (out, out + ".prev", out + ".contained"):
        if os.path.exists(leftover):
            os.remove(leftover)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
