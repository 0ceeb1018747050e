"""Dev helper (GPU box): per-generation kernel times on small scenes (RTX_AMD_GEN_TIMING=1)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import scenes
d = tempfile.mkdtemp()
paths = scenes.write_unit_scenes(d)
for name in sys.argv[1:] or ['tiny', 'plates_small']:
    hs = ra.HostScene(paths[name], use_bvh=True)
    ds = ra.DeviceScene(hs.desc)
    cam, _ = hs.camera(0)
    ds.render(cam)
    print('==', name, file=sys.stderr); sys.stderr.flush()
    ds.render(cam)
