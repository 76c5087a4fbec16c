#!/usr/bin/env python3
"""Where the device idles inside the timed steps of a rocprofv3 --kernel-trace: the union of all kernels' (and blits') busy intervals, the gaps between them by the
kernel that ended last in front of each gap.  usage: idle_gaps.py <dir with *kernel_trace.csv [*memory_copy_trace.csv]> [min gap us, default 20]"""
import csv, glob, os, re, sys, collections
def short(n):
    m = re.search(r"wtz_kernel_\w+<(K_\w+)", n)
    if m: return m.group(1)
    m = re.search(r"wtz_kernel_(\w+?)(?:I[L0-9]|<|$)", n)
    if m: return m.group(1)
    m = re.search(r"rocprim|hipcub", n)
    return "rocprim" if m else n[:40]
def main():
    d = sys.argv[1]; mingap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 and sys.argv[2][0] != "-" else 20e3
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r.get("Kernel_Name", ""))))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy"))
    if not ev: print("no trace under", d); return
    ev.sort()
    # the LAST step only (the first holds one-time page-locked allocations, the process start the FASTA load): a step begins with the k-mer index build's first kernel
    starts = [e[0] for e in ev if e[2] == "K_kcount"]
    if starts and "--all" not in sys.argv:
        cut = max(starts); ev = [e for e in ev if e[0] >= cut]
        print("# last step only (from the last K_kcount on); --all for the whole trace")
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    busy_end, last = ev[0][1], ev[0][2]; idle = 0; by = collections.Counter(); big = []
    for s, e, n in ev[1:]:
        if s > busy_end:
            g = s - busy_end
            if g >= mingap: idle += g; by[last] += g
            if g >= 1e6: big.append(((busy_end - t0) / 1e6, g / 1e6, last, n))
        if e > busy_end: busy_end, last = e, n
    ksum = collections.Counter()
    for s, e, n in ev: ksum[n] += e - s
    print("span %.1f ms, idle %.1f ms in gaps >= %.0f us (%.1f %%)" % ((t1 - t0) / 1e6, idle / 1e6, mingap / 1e3, 100.0 * idle / (t1 - t0)))
    print("idle ms by the kernel in front of the gap:", [(k, round(v / 1e6, 1)) for k, v in by.most_common(12)])
    print("kernel ms:", [(k, round(v / 1e6, 1)) for k, v in ksum.most_common(24)])
    print("gaps >= 1 ms (ms into the trace, gap ms, before, after):")
    for b in big[:80]: print("  %.1f  %.2f  %s -> %s" % b)
if __name__ == "__main__": main()
