#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: Gbp of candidate pairs aligned per second (wtzmo all-vs-all).

One "step" = one complete overlap phase of the drop-in wtzmo (k-mer index + z-mer index build, candidate search,
pair seeding / windows / chaining, banded SW, in-order commit, .ovl writing) over the synthetic E. coli-shape read set
(4.6 Mbp genome, 25x, mean 10 kb, 15 % error: BASELINE.json configs[1]) with the 2-bit reads ALREADY RESIDENT IN HBM.
The host driver runs in-process (libwtzmo_host.so = the C `wtzmo` main built as a shared object) so that exactly K
steps are bracketed by barrier + torch.cuda.synchronize() on both sides.

N > 1: one process per GPU (torchrun), the query set is sharded by the reference's own job striping (-P N -p rank,
wtzmo.c:1291,1314) with reads + index replicated in every GPU's HBM, no data-path collective; the overlap records are
gathered on rank 0 with RCCL (all_gather over xGMI) inside the timed region.  Total work is fixed -> "strong" scaling.

numerator  = sum over ranks and timed steps of (len(a)+len(b)) over pairs that entered pair alignment
value      = numerator / wall seconds of the K timed steps (max over ranks) / 1e9
roofline   = K-sw3 (shifting-band extension, the dominant DP stage; two concurrent kernels): DP cell updates exactly as
             the reference loops execute them x 12 int32 ops per cell / HIP-event time of the stage, against the int32 VALU peak
roofline_seed = seed lookup: algorithmic bytes (L/4 + 16 B per probe + 4 B per seed entry) / kernel time vs 8 TB/s
cpu_baseline  = the REAL reference `wtzmo -t <all cores>` (oracle/_ref, prebuilt) or the oracle port on a bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
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


def cpu_baseline(engine_argv, genome, coverage, seed, tmp):
    """Reference (or oracle port) timed on this host's cores on a bounded sample of the same workload shape."""
    ref = os.path.join(ROOT, "oracle", "_ref", "wtzmo_ref")
    ora = os.path.join(ROOT, "oracle", "wtzmo_oracle")
    fa = os.path.join(tmp, "cpu_sample_G%d_c%g_s%d.fa" % (genome, coverage, seed))
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
    ncpu = min(32, os.cpu_count() or 1)
    out = os.path.join(tmp, "cpu.ovl")
    if os.path.exists(ref):
        pairs = os.path.join(tmp, "cpu.pairs")
        t0 = time.time()
        subprocess.run([ref, "-t", str(ncpu), "-i", fa, "-fo", out, "-9", pairs] + engine_argv, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        bp = 0
        for line in open(pairs):
            a, b = line.split()
            bp += lens[a] + lens[b]
        return {"value": bp / dt / 1e9, "unit": "Gbp pair-bp/s", "cores": ncpu, "kind": "reference",
                "sample": "reference wtzmo -t %d on synthetic %d bp genome x%g (%d reads, %d bp), whole process wall %.2f s incl. FASTA load"
                          % (ncpu, genome, coverage, meta["reads"], meta["bases"], dt)}
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
    ap.add_argument("--genome", type=int, default=4600000, help="synthetic genome length (E. coli shape)")
    ap.add_argument("--coverage", type=float, default=25.0)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--engine", choices=["zmo", "dmo"], default="zmo")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--pool-gb", type=int, default=96)
    ap.add_argument("--cpu-genome", type=int, default=2300000, help="genome length of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

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

    eng = ZMO if a.engine == "zmo" else DMO
    out = os.path.join(tmp, "bench_r%d.ovl" % rank)
    stats = os.path.join(tmp, "bench_r%d.stats" % rank)
    W, K = a.warmup, a.steps
    argv = ["wtzmo", "--gpu", str(local), "-i", fa, "-fo", out, "--repeat", str(W + K), "--stats", stats, "--pool-gb", str(a.pool_gb)] + eng
    if a.max_batch:
        argv += ["--batch", str(a.max_batch)]
    argv += multigpu.stripe_argv(world, rank)

    host = C.CDLL(ge.HOSTLIB)
    T = {"t0": None, "t1": None, "gathered": 0}
    HOOK = C.CFUNCTYPE(None, C.c_int, C.c_int)

    def hook(rep, phase):
        if phase == 0 and rep == W:
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            T["t0"] = time.perf_counter()
        if phase == 1:
            if dist:   # RCCL gather of this step's overlap records (text lines; volume ~ MBs) inside the timed region
                blobs = multigpu.gather_records(dist, open(out, "rb").read(), "cuda")
                if rank == 0:
                    T["gathered"] = sum(len(b) for b in blobs)
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
        v = torch.tensor([float(pair_bp), float(n_pairs)], dtype=torch.float64, device="cuda")
        dist.all_reduce(v)
        pair_bp, n_pairs = int(v[0].item()), int(v[1].item())
        d = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        dt = float(d.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    value = pair_bp / dt / 1e9
    ms = {k: float(last[4 + i]) for i, k in enumerate(["index", "zindex", "candidates", "pairs", "winalign", "stitch"])}
    cells_shift = int(last[10])
    seed_bytes = int(last[13])
    ms_ext = float(last[15])
    res = {
        "metric": "Gbp of candidate pairs aligned/sec (wtzmo all-vs-all)",
        "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "E. coli-shape synthetic PacBio reads (BASELINE configs[1]): %d bp iid genome x%g, lognormal mean 10 kb, 15%% error (ins:del:sub 50:30:20)"
                               % (a.genome, a.coverage),
                   "reads": meta["reads"], "read_bases": meta["bases"], "engine": a.engine, "argv": " ".join(eng),
                   "parallelism": "1 GPU" if world == 1 else "query striping -P %d (reads + index replicated per GPU), RCCL gather of records" % world,
                   "parity": "records identical to `wtzmo -t 1%s` (tests/test_gpu_parity.py)" % ("" if world == 1 else " -P N -p rank` per rank")},
        "pairs_per_step": n_pairs // K, "pair_bp_per_step": pair_bp // K, "records_last_step": int(last[14]),
        "kernel_ms_last_step": dict(ms, ksw3_wave=ms_ext),
        "roofline": {"kernel": "K-sw3 shifting-band extension, DP rows in registers: wtz_kernel_extjobs_mw (four wavefronts per long job, side stream) "
                               "|| wtz_kernel_extjobs_reg (one wavefront per short job); time = HIP events around the pair of launches",
                     "bound": "valu_int32", "achieved": cells_shift * OPS_PER_CELL / (ms_ext * 1e-3) / 1e12 if ms_ext > 0 else None,
                     "peak": INT32_VALU_PEAK_TOPS, "unit": "Tint32op/s",
                     "frac": (cells_shift * OPS_PER_CELL / (ms_ext * 1e-3) / 1e12 / INT32_VALU_PEAK_TOPS) if ms_ext > 0 else None,
                     "cell_updates_per_s": cells_shift / (ms_ext * 1e-3) if ms_ext > 0 else None, "cells": cells_shift,
                     "backtrack_GBps": cells_shift / (ms_ext * 1e-3) / 1e9 if ms_ext > 0 else None, "traffic": None},
        "roofline_seed": {"kernel": "wtz_kernel_coop_tasks<K_candidates> (hzm seed lookup + candidate heap)", "bound": "hbm",
                          "achieved": seed_bytes / (ms["candidates"] * 1e-3) / 1e9 if ms["candidates"] > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (seed_bytes / (ms["candidates"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms["candidates"] > 0 else None, "algorithmic_bytes": seed_bytes, "traffic": None},
    }
    # HBM traffic of the two roofline kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, condensed by
    # tools/summarize_profiles.py into profiles/ (per launch = per-kernel total / dispatches; FETCH_SIZE doubled as the gfx950 guide says)
    try:
        import csv
        for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_%s_pmc_per_kernel.csv" % a.engine))):
            per_launch = float(row["hbm_bytes_est"]) / max(1, int(row["dispatches"]))
            if row["kernel"] in ("wtz_kernel_extjobs_reg", "wtz_kernel_extjobs_mw"):      # the two kernels of one K-sw3 stage launch
                res["roofline"]["traffic"] = (res["roofline"]["traffic"] or 0.0) + per_launch
                # executed wave-level VALU instructions of the stage (PMC pass of ONE step) x 2 issue cycles on a SIMD-32 / SIMD-cycles of the
                # live stage time: how much of the chip's VALU issue capacity the exact recurrence really occupies (the nominal 12 ops/cell
                # of SURVEY 8d undercounts it by ~8x)
                if ms_ext > 0 and row.get("SQ_INSTS_VALU"):
                    res["roofline"]["valu_issue_frac_pmc"] = (res["roofline"].get("valu_issue_frac_pmc") or 0.0) + float(row["SQ_INSTS_VALU"]) * 2.0 / (ms_ext * 1e-3 * 2.4e9 * 256 * 4)
                res["roofline"]["traffic_note"] = "HBM bytes per stage launch (both kernels) from profiles/r01_%s_pmc_per_kernel.csv (separate --pmc passes)" % a.engine
            if row["kernel"] == "K_candidates":
                res["roofline_seed"]["traffic"] = per_launch
    except Exception:
        pass
    if world > 1:
        res["gathered_record_bytes"] = T["gathered"]
    if world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(eng, a.cpu_genome, a.coverage, a.seed + 1000, tmp)
        except Exception as e:      # the baseline is reported, never required
            res["cpu_baseline"] = {"value": None, "error": str(e)}
    print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
