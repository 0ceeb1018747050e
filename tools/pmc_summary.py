#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel name.  Usage: tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k, c in tot.items():
    print(k, "dispatches(last pass):", len(calls[k]))
    for n, v in sorted(c.items()):
        print(f"   {n:40s} {v:.6g}")
