#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc.sh into profiles/<round>_<workload>_pmc.json: the
hardware-counter figures of the trace kernel PER KERNEL VARIANT (k_trace<COUNT, WIDE, FAST, NEST>), so
that the counters-on variant of the untimed counting pass never mixes with the timed one; bench.py
attaches the timed variant's figures to its roofline object.
Usage: tools/pmc_to_json.py <pmc dir> <workload> <out.json> [build id]

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
build = sys.argv[4] if len(sys.argv) > 4 else "?"
SIMDS = 256 * 4
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace(" ", "")
        if "k_trace<" not in name:
            continue
        name = name.split("(")[0].replace("voidrtxdev::", "")
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
if not tot:
    sys.exit("no k_trace dispatches found")


def figures(t, dsp):
    g = lambda k: t.get(k, 0.0)
    n = lambda k: max(len(dsp.get(k, ())), 1)
    chip_cycles = g("GRBM_GUI_ACTIVE") / 8.0                # busy cycles of the chip over the counted dispatches
    simd_cycles = chip_cycles * SIMDS
    fetch = 2.0 * g("FETCH_SIZE") * 1024.0 / n("FETCH_SIZE")
    write = g("WRITE_SIZE") * 1024.0 / n("WRITE_SIZE")
    return {
        "dispatches_per_pass": max(len(v) for v in dsp.values()),
        # fraction of the waves' resident time spent waiting for anything (both in quad-cycles)
        "wait_fraction": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
        "issue_fraction": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
        # VALU issue utilisation: quad-cycles with a VALU instruction in flight x 4, over SIMD-cycles of the busy chip
        "valu_busy_fraction": 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles if simd_cycles else None,
        # lanes enabled per VALU instruction (64 = every lane of the wave does work)
        "valu_lanes_per_instruction": g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") if g("SQ_ACTIVE_INST_VALU") else None,
        # resident waves per SIMD while the chip was busy: wave quad-cycles x 4 over SIMD-cycles
        "waves_per_simd_resident": 4.0 * g("SQ_WAVE_CYCLES") / simd_cycles if simd_cycles else None,
        "valu_instructions_per_launch": g("SQ_INSTS_VALU") / n("SQ_INSTS_VALU"),
        "salu_instructions_per_launch": g("SQ_INSTS_SALU") / n("SQ_INSTS_SALU"),
        "vmem_instructions_per_launch": g("SQ_INSTS_VMEM") / n("SQ_INSTS_VMEM"),
        "l2_hit_rate": g("TCC_HIT_sum") / g("TCC_REQ_sum") if g("TCC_REQ_sum") else None,
        # HBM-side bytes per launch: 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes) + WRITE_SIZE
        "hbm_fetch_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
        "counters": {k: v for k, v in sorted(t.items())},
    }


variants = {name: figures(t, disp[name]) for name, t in sorted(tot.items())}
# the timed variant: the one without work counters (k_trace<false, ...>) that the chip spent most time in
timed = max((n for n in variants if n.startswith("k_trace<false")), key=lambda n: tot[n].get("GRBM_GUI_ACTIVE", 0.0), default=None)
res = {"workload": workload, "build": build, "timed_variant": timed,
       "source": "rocprofv3 --pmc, one run per counter group (tools/pmc.sh); figures per kernel variant, averaged or summed over "
                 "that variant's dispatches only",
       "variants": variants}
json.dump(res, open(out, "w"), indent=1)
for name, v in variants.items():
    print(name, json.dumps({k: x for k, x in v.items() if k != "counters"}, indent=1))
print("timed variant:", timed)
