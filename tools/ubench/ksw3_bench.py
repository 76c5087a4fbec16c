#!/usr/bin/env python3
"""K-sw3 in isolation: a seeded sample of the extension jobs of one configs[2] zmo step (geometry dumped by `WTZ_PROFILE_PAIR=1 WTZ_EXT_DUMP=... bench.py`,
40 000 of 984 153 jobs: tools/ubench/ksw3_jobs_yeast100.npz) on synthetic homologous sequences (two 15 %-error copies of one random segment per job,
aligned from their common start - what an end extension sees), run through the device forms of kswx_extend_align_shift_core via the test-only ABI entry
wtz_test_dp:  1 = round-4 one-wave register kernel, 2 = four-wave kernel, 5 = one-wave kernel in the anti-diagonal frame (round 5), 6 = the frame form on four
wavefronts (round 6), 0 = the product's dispatch.
Every form's results are compared with form 1's (itself pinned to the reference's vectors by tests/test_gpu_dp_forms.py) - all fields and every CIGAR word.

  python tools/ubench/ksw3_bench.py [--forms 1,5,0] [--jobs 40000] [--reps 3]
prints one JSON line per form: ms per pass (HIP events around the launches), G cells/s (cells as the reference loops execute them), fraction of the int32 roof.
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from smartdenovo_amd import hipabi, synth  # noqa: E402


def geometry(qlen, tlen, W=800):
    w = np.minimum(W, np.maximum(qlen, tlen))
    ql = np.where(qlen < tlen, qlen, np.where(tlen + w < qlen, tlen + w, qlen))
    tl = np.where(qlen < tlen, np.where(qlen + w < tlen, qlen + w, tlen), tlen)
    return ql, tl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forms", default="1,5,0"); ap.add_argument("--jobs", type=int, default=40000); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=7); ap.add_argument("--pool-gb", type=int, default=128); ap.add_argument("--chunk", type=int, default=40000)
    ap.add_argument("--no-compare", action="store_true", help="diagnostic library builds (no trace, no traceback ...) give wrong results by design")
    a = ap.parse_args()
    d = np.load(os.path.join(ROOT, "tools", "ubench", "ksw3_jobs_yeast100.npz"))
    n = min(a.jobs, d["qlen"].size)
    qlen, tlen, init = d["qlen"][:n].astype(np.int64), d["tlen"][:n].astype(np.int64), d["init"][:n]
    # longest first, as run_extjobs orders a launch (a forced form gets no order list: the problem array itself is the order, so a pass is not its tail)
    o = np.argsort(-np.minimum(qlen, tlen), kind="stable"); qlen, tlen, init = qlen[o], tlen[o], init[o]
    ql, tl = geometry(qlen, tlen)
    rng = np.random.Generator(np.random.PCG64(a.seed))
    t0 = time.time()
    seqs = []
    for i in range(n):
        L = int(max(ql[i], tl[i]) * 1.15) + 64
        seg = rng.integers(0, 4, size=L, dtype=np.uint8)
        q = synth._mutate(seg, 0.15, rng); t = synth._mutate(seg, 0.15, rng)
        need_q, need_t = int(ql[i]), int(tl[i])
        if q.size < need_q: q = np.concatenate([q, rng.integers(0, 4, size=need_q - q.size, dtype=np.uint8)])
        if t.size < need_t: t = np.concatenate([t, rng.integers(0, 4, size=need_t - t.size, dtype=np.uint8)])
        seqs.append(q[:need_q]); seqs.append(t[:need_t])
    prob = np.zeros(n, dtype=hipabi.DP_PROBLEM)
    for i in range(n):
        prob[i] = (2 * i, 2 * i + 1, 0, 0, 0, 0, 1, 1, min(int(qlen[i]), seqs[2 * i].size), min(int(tlen[i]), seqs[2 * i + 1].size), int(init[i]), -800)
    # the geometry clips the longer side to the shorter + W, so the views above hold every base the DP reads; keep the ORIGINAL lengths where they fit
    for i in range(n):
        prob[i]["q_len"] = int(qlen[i]) if int(qlen[i]) <= seqs[2 * i].size else seqs[2 * i].size
        prob[i]["t_len"] = int(tlen[i]) if int(tlen[i]) <= seqs[2 * i + 1].size else seqs[2 * i + 1].size
    print("# %d jobs, %.1f Mbases, generated in %.0f s" % (n, sum(s.size for s in seqs) / 1e6, time.time() - t0), file=sys.stderr)
    ctx = hipabi.Context(hipabi.Params.defaults(), pool_bytes=a.pool_gb << 30)
    ctx.upload(*hipabi.pack_reads(seqs))
    ref = None
    peak = 256 * 4 * 32 * 2.4e9
    try:
        for form in [int(x) for x in a.forms.split(",")]:
            best = None
            for r in range(a.reps):
                ctx.reset_counters()
                outs, cigs = [], []
                for c0 in range(0, n, a.chunk):      # a forced form has no launch groups: keep the traces of one call inside the transient pool
                    o, cg = ctx.test_dp(hipabi.DP_SHIFT, form, prob[c0:c0 + a.chunk], cigar_cap=1 << 25)
                    outs.append(o); cigs.extend(cg)
                out = np.concatenate(outs)
                ms = ctx.counters().ms_ext
                best = ms if best is None else min(best, ms)
            cells = int(out["cells"].sum()); answered = int((out["form_used"] != 0).sum())
            line = {"form": form, "jobs": n, "answered": answered, "ms": round(best, 3), "cells": cells, "Gcells_per_s": round(cells / best / 1e6, 1),
                    "int32_roof_frac_at_12_ops": round(cells * 12 / (best * 1e-3) / peak, 4)}
            if ref is None:
                ref = (out, cigs, form)
            elif not a.no_compare:
                ro, rc, rf = ref
                both = (out["form_used"] != 0) & (ro["form_used"] != 0)
                bad = 0
                for f in ("score", "tb", "te", "qb", "qe", "aln", "mat", "mis", "ins", "del", "cigar_len", "cells"):
                    bad += int((out[f][both] != ro[f][both]).sum())
                for k in np.nonzero(both)[0]:
                    if cigs[k].size != rc[k].size or (cigs[k] != rc[k]).any():
                        bad += 1
                line["differs_from_form_%d" % rf] = bad; line["compared"] = int(both.sum())
            print(json.dumps(line), flush=True)
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
