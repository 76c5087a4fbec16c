#!/usr/bin/env python3
"""Regenerates the "Round-5 numbers" block of DESIGN.md §6 from the committed measurement files under profiles/ (bench lines, rocprofv3 kernel statistics,
PMC per kernel): python tools/design_numbers.py [--write].  Nothing in it is typed by hand."""
import csv, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); P = os.path.join(R, "profiles")
J = lambda n: json.loads(open(os.path.join(P, n)).read().strip().split("\n")[-1])
z, d, e, ed, fly = J("r05_bench_driver_cmd.json"), J("r05_bench_dmo.json"), J("r05_bench_ecoli_zmo.json"), J("r05_bench_ecoli_dmo.json"), J("r05_bench_fly70.json")
def cpu(x):
    c = x.get("cpu_baseline")
    if not c: return "–"
    m = re.search(r"overlap phase ([\d.]+) s", c["sample"]); a = x.get("cpu_baseline_all_cores")
    t = "`-t 32` on the bench input: %s s (%.3f Gbp/s) → ×%.0f" % (m.group(1), c["value"], x["value"] / c["value"])
    return t + ("; `-t 256` on a same-shape sample: %.3f Gbp/s" % a["value"] if a else "")
def pmc(path):
    r = {}
    for row in csv.DictReader(open(os.path.join(P, path))):
        r[row["kernel"]] = r.get(row["kernel"], 0) + float(row["hbm_bytes_est"])
    return r
def stats(path, steps):
    rows = list(csv.DictReader(open(os.path.join(P, path))))
    # the number of steps the trace covers comes from the file itself where it can: the z-mer index is built once per step (K_zrun: one launch per build at configs[2])
    for row in rows:
        if row["kernel"] == "K_zrun": steps = int(row["calls"])
    r = {}
    for row in rows:
        r[row["kernel"]] = r.get(row["kernel"], 0) + float(row["total_ms"]) / steps
    return r
pz, pd = pmc("r05_yeast100_zmo_pmc_per_kernel.csv"), pmc("r05_yeast100_dmo_pmc_per_kernel.csv")
sz, sd = stats("r05_yeast100_zmo_kernel_stats.csv", 4), stats("r05_yeast100_dmo_kernel_stats.csv", 2)
gb = lambda x: "%.0f GB" % (x / 1e9)
L = []
L.append("**Round-5 numbers** (MI355X; `profiles/r05_*`, made by `tools/gpu_r05_final.sh` on the last kernels of the round, this block by `tools/design_numbers.py`; every line's md5 == reference):")
L.append("")
L.append("| workload / engine | step | value | end of round 4 | reference on the same host (2×EPYC 9575F) |")
L.append("|---|---|---|---|---|")
L.append("| configs[2] zmo (`-k 16 -s 200 -m 0.6`), %s records, %.1f Gbp of pairs | **%.3f s** (%d steps) | **%.2f Gbp/s** | 2.895 s / 5.84 | %s |" % (format(z["records_last_step"], ",").replace(",", " "), z["pair_bp_per_step"] / 1e9, z["ms_per_step"] / 1e3, z["steps"], z["value"], cpu(z)))
L.append("| configs[2] dmo (`-U -1 -m 0.1 -A 1000 -Z 16`), %s records, %.1f Gbp | **%.3f s** | **%.2f Gbp/s** | 6.156 s / 13.35 | %s |" % (format(d["records_last_step"], ",").replace(",", " "), d["pair_bp_per_step"] / 1e9, d["ms_per_step"] / 1e3, d["value"], cpu(d)))
L.append("| configs[1] zmo, %s records | %.3f s | %.2f Gbp/s | 0.322 s / 2.99 | %s |" % (format(e["records_last_step"], ",").replace(",", " "), e["ms_per_step"] / 1e3, e["value"], cpu(e)))
L.append("| configs[1] dmo, %s records | %.3f s | %.2f Gbp/s | 0.249 s / 7.84 | – |" % (format(ed["records_last_step"], ",").replace(",", " "), ed["ms_per_step"] / 1e3, ed["value"]))
L.append("| configs[3] shape (9.8 Gbp of reads, all-reads z-index beside a pool sized from the input), %s records, %.0f Gbp | %.1f s | %.2f Gbp/s | 30.7 s / 4.01 | no whole-job reference (days); parity on the `-P 128` stripe |" % (format(fly["records_last_step"], ",").replace(",", " "), fly["pair_bp_per_step"] / 1e9, fly["ms_per_step"] / 1e3, fly["value"]))
L.append("")
L.append("| line | kernel(s) | cells (bytes) per step | kernel ms | frac | PMC traffic per step |")
L.append("|---|---|---|---|---|---|")
r = z["roofline"]; L.append("| `roofline` K-sw3 | `wtz_kernel_stitch_ext_fr` (+ `wtz_kernel_extjobs_fr` where the fused launch declines) | %.1f G cells (= trace bytes) | %.0f | **%.4f** of 78.6 Tint32op/s (%.0f G cells/s) | %s |" % (r["cells_per_step"] / 1e9, r["kernel_ms_per_step"], r["frac"], r["cell_updates_per_s"] / 1e9, gb(pz.get("wtz_kernel_stitch_ext_fr", 0) + pz.get("wtz_kernel_extjobs_fr", 0) + pz.get("wtz_kernel_extjobs", 0))))
r = z["roofline_sw1"]; L.append("| `roofline_sw1` K-sw1 | `K_lplan` → `K_ldp` → `K_ltb` → `K_lfold` (+ `K_winalign`) | %.1f G cells | %.0f | **%.4f** (%.0f G cells/s) | %s |" % (r["cells_per_step"] / 1e9, r["kernel_ms_per_step"], r["frac"], r["cell_updates_per_s"] / 1e9, gb(sum(pz.get(k, 0) for k in ("K_lplan", "K_ldp", "K_ltb", "K_lfold", "K_winalign")))))
r = z["roofline_sw2"]; L.append("| `roofline_sw2` K-sw2 | `K_gplan` → `K_gdp` → `K_gtb`, `K_gap` | %.1f G cells | %.0f | **%.4f** (%.0f G cells/s) | %s |" % (r["cells_per_step"] / 1e9, r["kernel_ms_per_step"], r["frac"], r["cell_updates_per_s"] / 1e9, gb(sum(pz.get(k, 0) for k in ("K_gplan", "K_gdp", "K_gtb", "K_gap")))))
r = z["roofline_zmer"]; L.append("| `roofline_zmer` zmo | `K_pair` | %.1f GB | %.0f | %.4f of 8 TB/s | %s |" % (r["algorithmic_bytes_per_step"] / 1e9, r["kernel_ms_per_step"], r["frac"], gb(pz["K_pair"])))
r = d["roofline_zmer"]; L.append("| `roofline_zmer` dmo | `K_pair_dm` (+ `K_pair_big`) | %.1f GB | %.0f | %.4f of 8 TB/s | %s |" % (r["algorithmic_bytes_per_step"] / 1e9, r["kernel_ms_per_step"], r["frac"], gb(pd.get("K_pair_dm", 0) + pd.get("K_pair_big", 0))))
r = z["roofline_seed"]; r2 = d["roofline_seed"]; L.append("| `roofline_seed` zmo / dmo | `K_candidates_wg` | %.1f GB / %.1f GB | %.0f / %.0f | %.4f / %.4f of 8 TB/s | %s / %s |" % (r["algorithmic_bytes_per_step"] / 1e9, r2["algorithmic_bytes_per_step"] / 1e9, r["kernel_ms_per_step"], r2["kernel_ms_per_step"], r["frac"], r2["frac"], gb(pz["K_candidates_wg"]), gb(pd["K_candidates_wg"])))
r = z["roofline_ingest"]; L.append("| `roofline_ingest` | `wtz_kernel_pack_ascii` (load time, outside the steps) | %.2f GB | %.3f | **%.2f** of 8 TB/s | – |" % (r["algorithmic_bytes"] / 1e9, r["kernel_ms"], r["frac"]))
L.append("")
L.append("PMC traffic = 2 × FETCH_SIZE + WRITE_SIZE (the gfx950 units and corrections of the guide's rocprofv3 section), separate `--pmc` passes of a `--steps 1` run, `profiles/r05_yeast100_{zmo,dmo}_pmc_per_kernel.csv` (+ `.meta.json`: the kernel-source hash they were measured on).")
L.append("")
k = z["kernel_ms_last_step"]; g = lambda n: sz.get(n, 0)
L.append("Where a configs[2] zmo step goes (rocprofv3, `profiles/r05_yeast100_zmo_kernel_stats.csv`, per step): K-sw3 %.0f ms (`wtz_kernel_stitch_ext_fr` %.0f + `wtz_kernel_extjobs_fr` %.0f ms), `K_pair` %.0f, K-sw1 stage %.0f (`K_ldp` %.0f, `K_ltb` %.0f, `K_lplan` %.0f, `K_lfold` %.0f, `K_winalign` %.0f), K-sw2 %.0f (`K_gap` %.0f, `K_gdp` %.0f, `K_gplan` %.0f, `K_gtb` %.0f), stitch glue %.0f (`K_stitch_mid` %.0f, `K_stitch_left` %.0f), z-index %.0f, seed lookup %.0f, k-mer index %.0f; rank-0 commit and the writer threads run beside the device stages."
         % (k["ksw3_wave"], g("wtz_kernel_stitch_ext_fr"), g("wtz_kernel_extjobs_fr"), k["pairs"], k["winalign"], g("K_ldp"), g("K_ltb"), g("K_lplan"), g("K_lfold"), g("K_winalign"), k["ksw2_gap"], g("K_gap"), g("K_gdp"), g("K_gplan"), g("K_gtb"),
            g("K_stitch_mid") + g("K_stitch_left") + g("K_stitch_fin") + g("K_cigar_text"), g("K_stitch_mid"), g("K_stitch_left"), k["zindex"], k["candidates"], k["index"]))
k = d["kernel_ms_last_step"]
L.append("dmo: `K_pair_dm` %.0f ms of %.0f (tiers by LDS need + `K_pair_big` %.0f ms), seed lookup %.0f (all 116 541 reads are queried: no masking in this engine), z-index %.0f, k-mer index %.0f." % (k["pairs"], d["ms_per_step"], sd.get("K_pair_big", 0), k["candidates"], k["zindex"], k["index"]))
block = "\n".join(L) + "\n"
if "--write" in sys.argv:
    p = os.path.join(R, "DESIGN.md"); s = open(p).read()
    i0 = s.index("**Round-5 numbers**"); i1 = s.index("What bounds them (details and the experiments")
    open(p, "w").write(s[:i0] + block + "\n" + s[i1:])
else:
    print(block)
