"""The command-line host on the GPU: the reference's flags and output files
(raytracer.cpp:173-282), and the animation pipeline behind several scene files (geometry kept
from frame to frame, pictures written in the background)."""
import json
import os
import subprocess

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

CLI = os.path.join(ck.ROOT, "raytracer_amd", "lib", "raytracer_amd")


def read_p3(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int32).reshape(h, w, 3).astype(np.uint8)


def run_cli(args):
    r = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_animation_pipeline_equals_frame_by_frame_runs(tmp_path):
    d = str(tmp_path)
    n = 5
    frames = [scenes.scene_david_frame(d, f, n_frames=12, faces=5000, res=(96, 80)) for f in range(n)]
    # frame 3 is a different scene (another material): the pipeline must reload there
    doc = json.load(open(frames[3]))
    doc["Scene"]["Materials"]["Material"][1]["DiffuseReflectance"] = "0.2 0.7 0.3"
    json.dump(doc, open(frames[3], "w"))
    together, single = os.path.join(d, "together"), os.path.join(d, "single")
    os.makedirs(together); os.makedirs(single)
    out = run_cli(frames + ["--out", together, "--png", "--inflight", "1"])
    assert out.count("geometry kept") == 2                       # frames 1, 2 (4 reloads after 3)
    for f in frames:
        run_cli([f, "--out", single, "--quiet"])
    for k in range(n):
        stem = f"davids_camera_{k:03d}"
        a, b = read_p3(os.path.join(together, stem + ".ppm")), read_p3(os.path.join(single, stem + ".ppm"))
        same, mx = ck.image_diff(a, b)
        assert mx == 0 and same == 1.0, (k, same, mx)          # (sums are order-independent: same bytes)
        res = json.load(open(os.path.join(together, stem + "_results.json")))
        assert res["geometryReused"] == (k in (1, 2))
        assert res["useBVH"] is True and res["enableBackFaceCulling"] is False
        assert os.path.getsize(os.path.join(together, stem + ".png")) > 0
        # and the picture is the reference's for that file (CPU restatement), to the parity bar
        hs = rt.HostScene(frames[k])
        cam, _ = hs.camera(0)
        o_rgb, _, o_ctr, _ = ck.oracle_render(hs.desc, cam, want_hits=False)
        hs.close()
        same, mx = ck.image_diff(a, o_rgb)
        assert mx <= 2 and same >= 0.999, (k, same, mx)
        assert res["rays"] == o_ctr["rays"] and res["nGpus"] == 1 and res["MraysPerS"] > 0
    assert not np.array_equal(read_p3(os.path.join(together, "davids_camera_000.ppm")),
                              read_p3(os.path.join(together, "davids_camera_001.ppm")))


def test_sharded_animation_covers_every_frame_once(tmp_path):
    d = str(tmp_path)
    frames = [scenes.scene_david_frame(d, f, n_frames=12, faces=2000, res=(48, 40)) for f in range(5)]
    outs = []
    for r in range(2):
        o = os.path.join(d, f"rank{r}")
        os.makedirs(o)
        run_cli(frames + ["--out", o, "--shard", f"{r}/2", "--quiet"])
        outs.append(sorted(x for x in os.listdir(o) if x.endswith(".ppm")))
    assert outs[0] == [f"davids_camera_{k:03d}.ppm" for k in (0, 2, 4)]
    assert outs[1] == [f"davids_camera_{k:03d}.ppm" for k in (1, 3)]


def test_cli_reports_bad_input(tmp_path):
    r = subprocess.run([CLI, os.path.join(str(tmp_path), "missing.json"), "--out", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode != 0 and "cannot be loaded" in r.stderr


def test_gpu_build_flag_gives_the_same_pictures(tmp_path):
    d = str(tmp_path)
    frames = [scenes.scene_david_frame(d, f, n_frames=12, faces=5000, res=(96, 80)) for f in range(2)]
    host_dir, dev_dir = os.path.join(d, "host"), os.path.join(d, "dev")
    os.makedirs(host_dir); os.makedirs(dev_dir)
    run_cli(frames + ["--out", host_dir, "--quiet"])
    out = run_cli(frames + ["--out", dev_dir, "--gpu-build", "--inflight", "1"])
    assert out.count("geometry kept") == 1
    for k in range(2):
        a = read_p3(os.path.join(host_dir, f"davids_camera_{k:03d}.ppm"))
        b = read_p3(os.path.join(dev_dir, f"davids_camera_{k:03d}.ppm"))
        same, mx = ck.image_diff(a, b)
        assert mx == 0 and same == 1.0


def test_frame_split_over_devices_and_resolution_override(tmp_path):
    """--devices: the frame is split into interleaved tiles over several device scenes (here two
    tile streams on the one GPU) and assembled by the host; --resolution overrides the scene
    file's ImageResolution.  Both against the CPU restatement."""
    d = str(tmp_path)
    path = scenes.scene_grass(d, n_instances=200, res=(100, 60), name="cli_grass")
    one, two = os.path.join(d, "one"), os.path.join(d, "two")
    os.makedirs(one); os.makedirs(two)
    run_cli([path, "--out", one, "--quiet"])
    run_cli([path, "--out", two, "--quiet", "--devices", "0,0", "--tile", "16x8"])
    a, b = read_p3(os.path.join(one, "cli_grass.ppm")), read_p3(os.path.join(two, "cli_grass.ppm"))
    same, mx = ck.image_diff(a, b)
    assert mx == 0 and same == 1.0, (same, mx)
    hs = rt.HostScene(path)
    cam, _ = hs.camera(0)
    o_rgb, _, o_ctr, _ = ck.oracle_render(hs.desc, cam, want_hits=False)
    same, mx = ck.image_diff(b, o_rgb)
    assert mx <= 2 and same >= 0.999, (same, mx)
    res = json.load(open(os.path.join(two, "cli_grass_results.json")))
    assert res["nGpus"] == 2 and res["rays"] == o_ctr["rays"]
    # another resolution, same view
    big = os.path.join(d, "big")
    os.makedirs(big)
    run_cli([path, "--out", big, "--quiet", "--resolution", "150x90", "--gpus", "1"])
    c = read_p3(os.path.join(big, "cli_grass.ppm"))
    assert c.shape == (90, 150, 3)
    cam.width, cam.height = 150, 90
    o_big, _, _, _ = ck.oracle_render(hs.desc, cam, want_hits=False)
    same, mx = ck.image_diff(c, o_big)
    assert mx <= 2 and same >= 0.999, (same, mx)
    hs.close()


def test_frames_in_flight_give_the_same_files(tmp_path):
    """--inflight K: K pipelines (own loaded scene, device copy, host thread) take the frames in turn;
    the pictures are the same bytes as one pipeline's."""
    d = str(tmp_path)
    frames = [scenes.scene_david_frame(d, f, n_frames=12, faces=3000, res=(64, 48)) for f in range(7)]
    outs = {}
    for k in (1, 3):
        o = os.path.join(d, f"inflight{k}")
        os.makedirs(o)
        text = run_cli(frames + ["--out", o, "--inflight", str(k)])
        assert text.count("rendered in") == 7
        outs[k] = o
    for f in range(7):
        name = f"davids_camera_{f:03d}.ppm"
        assert open(os.path.join(outs[1], name), "rb").read() == open(os.path.join(outs[3], name), "rb").read()
