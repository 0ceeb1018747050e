#!/bin/bash
# Dev helper (no GPU needed): the host pipeline (scene parser, PLY reader, BVH builder, CLI) built
# with AddressSanitizer + UBSan as one executable and run over a few generated scenes and over
# truncated / corrupted copies of a scene file and of a PLY file.  Without a GPU every run ends at
# "no HIP device available", after the whole host side has done its work.  Prints the number of runs
# in which a sanitizer reported something or the program died of a signal (expected: 0).
cd "$(dirname "$0")/.."
set -e
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -pthread \
    -o /tmp/rt_asan raytracer_amd/host/main.cpp raytracer_amd/host/host_scene.cpp \
    -Lraytracer_amd/lib -lrtx_amd -Wl,-rpath,$PWD/raytracer_amd/lib -Wl,-rpath-link,/opt/rocm/lib
python3 - <<'PY'
import glob, os, random, shutil, subprocess, sys
sys.path.insert(0, os.getcwd())
from raytracer_amd import scenes
d = "/tmp/asan_scenes"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
paths = [scenes.scene_tiny(d), scenes.scene_grass(d, n_instances=50, res=(32, 18), name="g50")]
paths += [scenes.scene_random(d, seed) for seed in (1, 2, 3, 110)]

def run(scene):
    r = subprocess.run(["/tmp/rt_asan", scene, "--quiet"], capture_output=True, text=True, timeout=120)
    out = r.stdout + r.stderr
    return "ERROR: AddressSanitizer" in out or "runtime error" in out or r.returncode < 0 or r.returncode > 2, out

bad = 0
for p in paths:
    b, out = run(p)
    bad += b
random.seed(5)
src = open(paths[0], "rb").read()
ply = glob.glob(d + "/*.ply")[0]
ply_src = open(ply, "rb").read()
for i in range(60):
    data = bytearray(src[:random.randrange(1, len(src))] if i < 30 else src)
    if i >= 30:
        for _ in range(random.randrange(1, 6)):
            data[random.randrange(len(data))] = random.randrange(32, 127)
    open(d + "/fz.json", "wb").write(bytes(data))
    b, out = run(d + "/fz.json")
    bad += b
for i in range(40):
    data = bytearray(ply_src[:random.randrange(1, len(ply_src))] if i < 20 else ply_src)
    if i >= 20:
        for _ in range(random.randrange(1, 8)):
            data[random.randrange(len(data))] = random.randrange(256)
    open(ply, "wb").write(bytes(data))
    b, out = run(paths[1])
    bad += b
print("runs with a sanitizer report or a crash:", bad)
PY
