#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel_stats.csv, counter_collection.csv) into per-kernel tables small enough to commit under profiles/.

usage: summarize_profiles.py <dir with trace_*/ and pmc_*/ sub-directories> <out dir>
HBM traffic per kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: the counters are in KiB and FETCH_SIZE reports half of the
bytes of a wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section); the x2 is therefore an upper estimate for narrow accesses.
"""
import csv, glob, os, re, sys, collections

def short(name):
    m = re.search(r"wtz_kernel_\w+<(K_\w+)", name)
    if m:
        return m.group(1)
    for pat in (r"wtz_kernel_\w+", r"rocprim::(?:detail::)?\w+"):
        m = re.search(pat, name)
        if m:
            return m.group(0)
    return name[:80]

def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace_*", "**", "*kernel_stats.csv"), recursive=True):
        tag = os.path.basename(os.path.dirname(os.path.dirname(f))) if "trace_" not in os.path.basename(os.path.dirname(f)) else os.path.basename(os.path.dirname(f))
        tag = [p for p in f.split(os.sep) if p.startswith("trace_")][0]
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as o:
            w = csv.writer(o)
            w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.1f" % (float(r["AverageNs"]) / 1e3),
                            "%.1f" % (float(r["MinNs"]) / 1e3), "%.1f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if r["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVES") and key not in seen:
                seen.add(key); calls[(k, r["Counter_Name"])] += 1
    if agg:
        names = sorted({c for v in agg.values() for c in v})
        with open(os.path.join(dst, "pmc_per_kernel.csv"), "w") as o:
            w = csv.writer(o)
            w.writerow(["kernel", "dispatches"] + names + ["hbm_bytes_est"])
            for k in sorted(agg, key=lambda k: -agg[k].get("FETCH_SIZE", 0)):
                v = agg[k]
                hbm = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
                w.writerow([k, max(calls.get((k, "FETCH_SIZE"), 0), calls.get((k, "SQ_WAVES"), 0))] + ["%.0f" % v.get(c, 0) for c in names] + ["%.0f" % hbm])

if __name__ == "__main__":
    main()
