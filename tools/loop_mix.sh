#!/bin/bash
# Dev helper: instruction mix of the traversal loop (the innermost loop around the sixteen LDS-DMA
# loads) of one k_trace variant, from the device assembly.
# Usage: tools/loop_mix.sh [mangled-name fragment, default ILb0ELi1ELb0ELi0E] [extra -D flags]
cd "$(dirname "$0")/.."
frag=${1:-ILb0ELi1ELb0ELi0E}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-strict-aliasing \
  -munsafe-fp-atomics -fPIC -Iinclude "$@" -S --cuda-device-only -o /tmp/rtx_core.s raytracer_amd/csrc/rtx_core.hip 2>/dev/null
python3 - "$frag" <<'PY'
import re, sys, collections
frag = sys.argv[1]
lines = open("/tmp/rtx_core.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN6rtxdev7k_trace" + frag + r".*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
dma = [i for i, l in enumerate(body) if "global_load_lds_dwordx4" in l]
# the loop header named by the block that holds the first LDS-DMA load
hdr = None
for i in range(dma[0], 0, -1):
    m = re.search(r"in Loop: Header=(BB\d+_\d+)", body[i])
    if m:
        hdr = m.group(1)
        break
first = next(i for i, l in enumerate(body) if l.startswith("." + "L" + hdr + ":"))
last = max(i for i, l in enumerate(body) if "Header=" + hdr + " " in l)
# the last block runs to the next label
nxt = next(i for i in range(last + 1, len(body)) if body[i].startswith(".LBB"))
mix = collections.Counter()
for l in body[first:nxt]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    mix[re.sub(r"_e(32|64)$", "", op)] += 1
total = sum(mix.values())
valu = sum(v for k, v in mix.items() if k.startswith("v_"))
salu = sum(v for k, v in mix.items() if k.startswith("s_"))
print(f"loop {hdr}: lines {first}-{nxt} instructions {total} valu {valu} salu {salu} other {total - valu - salu}")
print("  " + "  ".join(f"{k}={v}" for k, v in mix.most_common(24)))
PY
