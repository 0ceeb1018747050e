"""Dev helper (GPU box): one frame of a workload with per-generation timing lines, through whichever build of the
library RTX_AMD_LIB names (tools/diag_build.sh: -DRTX_WAVE_TALLY=1..4 add wave lifetimes and section shares).
Usage: RTX_AMD_LIB=build_diag/librtx_amd_tally4.so python tools/tally.py <workload> [wide|counters|noscatter]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon_metal"
wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
counters = len(sys.argv) > 2 and sys.argv[2] == "counters"      # the counting kernel's own section clocks
path, _ = bench.make_workload(name, bench.scene_cache_dir())
hs = rt.HostScene(path)
ds = rt.DeviceScene(hs.desc, 0, wide_blas=wide)
if wide:
    ds.set_wide_blas(True)
ds.set_counters(counters)
cam, _ = hs.camera(0)
for _ in range(2):
    ds.render(cam)
extra = _abi.DEBUG_NO_SCATTER if (len(sys.argv) > 2 and sys.argv[2] == "noscatter") else 0
ds.set_debug(extra)
ds.render(cam)
ds.set_debug(_abi.DEBUG_GEN_TIMING | extra)
_, _, st = ds.render(cam)
print(f"{name}{' (eight-wide)' if wide else ''}: kernel {st['kernel_ms']:.3f} ms, trace {st['trace_ms']:.3f} ms, records {st['record_fetches']}", file=sys.stderr)
ds.close(); hs.close()
