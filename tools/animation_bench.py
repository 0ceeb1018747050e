#!/usr/bin/env python3
"""Dev helper (GPU box): the camera_around_david stand-in through the command-line host,
once as one pipeline over all frames and once as one process per frame (the reference's
runner.py pattern).  Usage: tools/animation_bench.py [n_frames]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes  # noqa: E402

CLI = os.path.join(ROOT, "raytracer_amd", "lib", "raytracer_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
d = tempfile.mkdtemp(prefix="rtx_anim_")
frames = [scenes.scene_david_frame(d, f, n_frames=360, faces=100_000, res=(800, 800)) for f in range(n)]
for mode in ("pipeline", "pipeline+png", "process per frame"):
    out = os.path.join(d, mode.replace(" ", "_").replace("+", "_"))
    os.makedirs(out)
    t = time.time()
    if mode.startswith("pipeline"):
        subprocess.check_call([CLI] + frames + ["--out", out, "--quiet"] + (["--png"] if "png" in mode else []))
    else:
        for f in frames:
            subprocess.check_call([CLI, f, "--out", out, "--quiet"])
    dt = time.time() - t
    print(f"{mode}: {n} frames in {dt:.2f} s = {1000 * dt / n:.1f} ms/frame", flush=True)
