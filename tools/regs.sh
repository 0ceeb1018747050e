#!/bin/bash
# Dev helper: register use, spills and LDS of every kernel in rtx_core.hip (extra -D flags may follow).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-strict-aliasing \
  -munsafe-fp-atomics -fPIC -Iinclude "$@" -c -o /tmp/rtx_core_regs.o raytracer_amd/csrc/rtx_core.hip \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur, vals = None, {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur, vals = m.group(1), {}
        continue
    m = re.search(r"remark: +([A-Za-z][^:]*): (\S+)", line)
    if m and cur:
        k = m.group(1).strip().split(" [")[0]
        vals[k] = m.group(2)
        if k.startswith("LDS"):
            name = re.sub(r"^_ZN6rtxdev\d+", "", cur)[:34]
            keep = ("TotalSGPRs", "VGPRs", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "LDS Size")
            print(name.ljust(36), "  ".join(f"{kk}={vals.get(kk)}" for kk in keep))
            cur = None
'
