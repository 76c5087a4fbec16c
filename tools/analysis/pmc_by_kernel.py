#!/usr/bin/env python3
"""counter totals per kernel name out of rocprofv3 --pmc csv output: pmc_by_kernel.py <dir>"""
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    print(k, {c: (round(v), n[(k, c)]) for c, v in agg[k].items()})
