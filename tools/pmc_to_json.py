#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc.sh into profiles/r02_<workload>_pmc.json: the
hardware-counter figures of the trace kernel (all k_trace variants that ran) that bench.py
attaches to its roofline object.  Usage: tools/pmc_to_json.py <pmc dir> <workload> <out.json>

Units: SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles) summed
over waves / SIMDs; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' busy cycles; FETCH_SIZE / WRITE_SIZE
are KiB, FETCH_SIZE tallying the 128-byte requests of 16-byte-per-lane reads at 64 bytes on gfx950
(hence doubled)."""
import collections
import csv
import glob
import json
import sys

d, workload, out = sys.argv[1:4]
SIMDS = 256 * 4
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
names = set()
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace(" ", "")
        if "k_trace<" not in name:
            continue
        names.add(name.split("(")[0].replace("voidrtxdev::", ""))
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
if not tot:
    sys.exit("no k_trace dispatches found")
g = lambda k: tot.get(k, 0.0)
chip_cycles = g("GRBM_GUI_ACTIVE") / 8.0                # busy cycles of the chip over the counted dispatches
simd_cycles = chip_cycles * SIMDS
res = {
    "workload": workload, "kernels": sorted(names),
    "source": "rocprofv3 --pmc, one run per counter group (tools/pmc.sh), summed over every k_trace dispatch of the run",
    "counters": {k: v for k, v in sorted(tot.items())},
    "dispatches_per_pass": {k: len(v) for k, v in sorted(disp.items())},
    # fraction of the waves' resident time spent waiting for anything (both in quad-cycles)
    "wait_fraction": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
    # VALU issue utilisation: quad-cycles with a VALU instruction in flight x 4, over SIMD-cycles of the busy chip
    "valu_busy_fraction": 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles if simd_cycles else None,
    "valu_busy_fraction_uncorrected": g("SQ_ACTIVE_INST_VALU") / simd_cycles if simd_cycles else None,
    # lanes enabled per VALU instruction (64 = every lane of the wave does work)
    "valu_lanes_per_instruction": g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") if g("SQ_ACTIVE_INST_VALU") else None,
    # resident waves per SIMD while the chip was busy: wave quad-cycles x 4 over SIMD-cycles
    "waves_per_simd_resident": 4.0 * g("SQ_WAVE_CYCLES") / simd_cycles if simd_cycles else None,
    "valu_instructions": g("SQ_INSTS_VALU"), "vmem_instructions": g("SQ_INSTS_VMEM"),
    "l2_hit_rate": g("TCC_HIT_sum") / g("TCC_REQ_sum") if g("TCC_REQ_sum") else None,
    "hbm_fetch_bytes": 2.0 * g("FETCH_SIZE") * 1024.0, "hbm_write_bytes": g("WRITE_SIZE") * 1024.0,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("counters", "dispatches_per_pass")}, indent=1))
