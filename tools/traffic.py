#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh) into profiles/traffic.json:
HBM-side bytes per launch of the trace kernel's main variant.  FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 bytes per lane) reads at 64 bytes, so it is
doubled; WRITE_SIZE is taken as is.  Usage: tools/traffic.py <pmc dir> <workload> <out.json>"""
import csv, glob, json, sys
d, workload, out = sys.argv[1:4]
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
n = {"FETCH_SIZE": set(), "WRITE_SIZE": set()}
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        # the main variant of the trace kernel (k_trace<false, 0, ...> or <false, 1, ...>), not the tiny-generation one
        name = r["Kernel_Name"].replace(" ", "")
        if not ("k_trace<false,0" in name or "k_trace<false,1" in name) or r["Counter_Name"] not in tot:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
lf, lw = len(n["FETCH_SIZE"]), len(n["WRITE_SIZE"])
if not lf or not lw:
    sys.exit("no dispatches of the main trace kernel with FETCH_SIZE and WRITE_SIZE found")
fetch = 2.0 * tot["FETCH_SIZE"] * 1024.0 / lf
write = tot["WRITE_SIZE"] * 1024.0 / lw
json.dump({"workload": workload, "kernel": "k_trace<false, 0|1> (main variant)", "hbm_bytes_per_launch": fetch + write,
           "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
           "launches_fetch_pass": lf, "launches_write_pass": lw,
           "note": "2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes) + WRITE_SIZE, KiB -> bytes, "
                   "separate --pmc passes, averaged over the launches of the pass"}, open(out, "w"), indent=1)
print(open(out).read())
