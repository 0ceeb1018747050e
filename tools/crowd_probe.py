import os, sys, tempfile
sys.path.insert(0, "/root/repo")
os.environ["RTX_AMD_GEN_TIMING"] = "1"
import raytracer_amd as rt
from raytracer_amd import scenes
d = tempfile.mkdtemp()
for seed in (0, 3, 8, 13):
    path = scenes.scene_random(d, seed, res=(96, 72), crowd=40 + 12 * (seed % 9))
    hs = rt.HostScene(path, True, bool(seed & 1)); ds = rt.DeviceScene(hs.desc, 0); cam, _ = hs.camera(0)
    print("==== seed", seed, file=sys.stderr, flush=True)
    ds.render(cam)
