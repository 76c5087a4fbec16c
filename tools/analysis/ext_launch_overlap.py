#!/usr/bin/env python3
"""K-sw3 launches of one step out of a rocprofv3 --kernel-trace: for every burst of extension kernels (launches that overlap or follow each other within 50 us)
the span from the first start to the last end, the busy time of each kernel in it and how much of the span more than one of them was running.
usage: ext_launch_overlap.py <dir with *kernel_trace.csv> [name filter, default 'extjobs|stitch_ext']"""
import csv, glob, os, re, sys
def main():
    d = sys.argv[1]; pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"extjobs|stitch_ext")
    fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not fs: print("no kernel trace under", d); return
    ev = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name") or r.get("Name") or ""
            if not pat.search(n): continue
            m = re.search(r"wtz_kernel_(\w+?)(?:<|I[L0-9]|$)", n); short = m.group(1) if m else n[:40]
            t = re.search(r"Li(\d+)ELi(\d+)E", n)
            if t: short += "<%s,%s>" % (t.group(1), t.group(2))
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
    ev.sort()
    bursts = []; cur = []
    for e in ev:
        if cur and e[0] > max(x[1] for x in cur) + 50000: bursts.append(cur); cur = []
        cur.append(e)
    if cur: bursts.append(cur)
    tot_span = tot_busy = 0.0
    for b in bursts:
        s0 = min(x[0] for x in b); s1 = max(x[1] for x in b); span = (s1 - s0) / 1e6
        pts = sorted([(x[0], 1) for x in b] + [(x[1], -1) for x in b]); depth = 0; last = s0; multi = 0
        for t, dlt in pts:
            if depth > 1: multi += t - last
            depth += dlt; last = t
        tot_span += span; tot_busy += sum(x[1] - x[0] for x in b) / 1e6
        print("span %7.2f ms  overlapped %6.2f ms | " % (span, multi / 1e6) + "  ".join("%s[%d wg] +%.2f..%.2f" % (x[2], x[3] // 64 if x[3] else 0, (x[0] - s0) / 1e6, (x[1] - s0) / 1e6) for x in b))
    print("# %d bursts, spans sum %.1f ms, kernel times sum %.1f ms" % (len(bursts), tot_span, tot_busy))
if __name__ == "__main__": main()
