#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one frame per GPU through rtx_render.

Metric: Mrays/s and ms/frame.  A ray is one closest-hit query the reference semantics require
(one intersect() call, utils.cpp:671, not counting calls whose result computePixelColor
discards); the per-frame ray count is taken from the device's own counters in an untimed pass
and is, by construction and by test, the count the CPU restatement produces.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--mode frames|tiles]

N > 1 without WORLD_SIZE in the environment: this process starts N ranks of itself (one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything touches a GPU), relays
rank 0's JSON line and exits with the first failing rank's code.  Under torchrun
(WORLD_SIZE set) it is one of the ranks.

Workloads and how they shard (the scene is replicated on every GPU, no ray leaves its GPU):
  dragon_metal (default), marching_dragons, metal_glass_plates, grass_desert, mode frames:
      the K timed steps are a K-frame turntable around the scene (frame 0 = the scene file's own
      camera); every rank renders each frame once, rank r starting r/N of the way round -- whole
      frames, no collective on the data path, the same work per GPU at every N (weak scaling).
  camera_around_david: the 360-frame sequence itself; rank r renders frames r, r+N, ... through
      the animation pipeline (HostScene.load_next keeps the geometry, update_lights); one step
      = one frame per rank (weak scaling).
  --mode tiles (grass_desert at 4K by default): ONE frame split into interleaved tiles, one
      gather to rank 0 over RCCL (raytracer_amd.distributed.gather_tiled_frame), strong scaling.
Unless --no-extras is given, the default run also measures the other two shardings after the
main measurement and reports them as "sequence" and "tiled_frame" inside the same line.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # HBM3E, spec
L2_PEAK_GBS = 34500.0        # 8 x 4 MiB L2, aggregate
L2_GATHER_GBS = 17800.0      # whole records gathered into LDS out of an XCD's L2: 16.8-18.8 TB/s measured chip-wide
L2_BYTES = 32 << 20          # 8 x 4 MiB
PROFILE_ROUND = "r03"        # the committed profiles this build's line quotes (profiles/<round>_<workload>_*.json)
FP64_VALU_PEAK_TF = 78.6     # vector FP64 (what the path computes in)
FP32_VALU_PEAK_TF = 157.3    # vector FP32
SEQUENCE_FRAMES = 360


# ---------------------------------------------------------------------------------------------
# Launching the ranks
# ---------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv, script=None, python=None, extra_env=None) -> int:
    """Start n ranks of `script` (default: this file) as fresh child processes and wait for
    them.  Rank 0's stdout is relayed to ours; the other ranks' stdout goes to stderr.  Returns 0
    when every rank succeeded, otherwise the first non-zero exit code seen (the remaining ranks
    are then terminated: they would wait in a collective forever).  The parent never touches a
    GPU, and no process that has is ever replaced by another program."""
    script = script or os.path.abspath(__file__)
    python = python or sys.executable
    port = free_port()
    procs = []
    for r in range(n):
        env = os.environ.copy()
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([python, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's line is read while the ranks run: a line longer than the pipe's buffer would otherwise
    # block rank 0 in its write, and with it every rank waiting for rank 0 in a collective
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending:
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    reader.join(timeout=30)
    out = b"".join(chunks).decode(errors="replace")
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def selftest_rank(args) -> int:
    """--selftest-ranks: what a rank does when only the launch path is being checked (no GPU):
    rendezvous over gloo, one all-reduce, rank 0 prints a line with the world size."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_fail_rank == rank:
        return 7
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    # the self-checks of a real N > 1 line, on stand-in data: every rank is seen by the collective,
    # per-rank step times are gathered, and a frame every rank "renders" hashes alike on all of them
    import numpy as np
    R = Rank.__new__(Rank)
    R.torch, R.world, R.rank, R.dist, R.red_dev, R.backend = torch, world, rank, (dist if world > 1 else None), "cpu", "gloo"
    frame = (np.arange(4096, dtype=np.uint32) * 2654435761 % 251).astype(np.uint8)
    if args.selftest_differ_rank == rank:
        frame[17] ^= 1
    check = multi_rank_check(R, frame, ms_per_step=1.0 + rank)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": float(t.item()),
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "multi_gpu": check}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------
# Work counts, algorithmic bytes and flops
# ---------------------------------------------------------------------------------------------
class OracleCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "primary_rays", "shadow_rays", "secondary_rays", "node_pops",
                                          "tri_tests", "world_box_tests", "ray_transforms", "transformed_hits",
                                          "shades", "tri_past_det", "tri_past_beta", "tri_past_gamma")]


def reference_bytes(c) -> int:
    """Algorithmic traffic of one frame in the reference's own in-memory record sizes: what the
    reference's traversal has to read per unit of work, whatever layout the device uses.
       72 B  one BVH node whose box is tested           (sizeof(BVHNode), bvh.h:50-61)
       88 B  one triangle test                          (4 B faceIndices + 12 B face + 3 x 24 B vertices)
       48 B  one world-space AABB test                  (AABB, bvh.h:11-16)
      128 B  one ray taken to object space              (one Matrix4x4, scene.h:123-128)
      384 B  one accepted back-transformed hit          (three Matrix4x4)
       96 B  per ray                                    (ray in + hit record out)"""
    return (72 * c["node_pops"] + 88 * c["tri_tests"] + 48 * c["world_box_tests"] + 128 * c["ray_transforms"]
            + 384 * c["transformed_hits"] + 96 * c["rays"])


def device_bytes(st, record_fetches=None, settled_early=0) -> int:
    """A frame in the device's own records.  With `record_fetches` (rtx_stats.record_fetches of the
    timed frames, counted by the trace kernel in every mode): 256 bytes per record the traversal
    fetched -- collapsed nodes, triangle pairs, instance set-up records -- plus 96 bytes of ray
    record in per queued ray (`settled_early` of the rays were not) and 128 bytes of hit record out
    per radiance ray.  Without it, the
    same from the work counters of a counters-on frame (every test of the reference carried out):
    256 bytes per BLAS or TLAS record visit, 128 per triangle test, world-box test and ray taken
    to object space."""
    radiance = st["rays"] - st["shadow_rays"]
    # (shadow rays the shading kernel settled where it would have emitted them were never queued:
    # rtx_stats.shadow_rays_settled_early of the timed frames)
    rays_io = 96 * (st["rays"] - st["primary_rays"] - int(settled_early)) + 128 * radiance
    if record_fetches is not None:
        return int(256 * record_fetches) + rays_io
    return (256 * (st["node_visits"] + st["tlas_node_visits"]) + 128 * st["tri_tests"] + 128 * st["world_box_tests"]
            + 128 * st["object_space_traversals"] + rays_io)


# FP64 operations of the device's own arithmetic, counted from raytracer_amd/csrc/rtx_kernels.h
# (adds, multiplies, min/max and compares count one each, a division or square root one):
#   box_enter_nopar: 6 sub + 6 mul + 6 min/max of pairs + 3 max (entry) + 2 min (exit) + 2 compares = 25;
#       a four-wide record visit tests four boxes = 100
#   tri_test: to the determinant guard 3 (o - a) + 9 (e2 x d) + 6 (det) + 1 = 19; then the division,
#       detB (6), beta and its compares = +10; r x d (9), detG (6), gamma, compares = +20;
#       e2 x r (9), detT (5), t, compares = +17 (66 for a test that reaches the end)
#   job set-up of a transformed object (xf_point 25, xf_dir 15, normalize_f32 10, 3 divisions) = 53, its box 25
#   k_shade: about 150 per shaded hit (normals, Fresnel, children) + 60 per light (shadow-ray set-up, Blinn-Phong)
def frame_flops(st, ref_counts) -> float:
    visits = st["node_visits"] + st["tlas_node_visits"]
    f = 100.0 * visits + 25.0 * st["world_box_tests"] + 53.0 * st["object_space_traversals"]
    if ref_counts is not None and ref_counts.get("tri_tests"):
        # stage counts of the reference's algorithm (same tests, same arithmetic), scaled to the device's test count
        k = st["tri_tests"] / ref_counts["tri_tests"]
        f += k * (19.0 * ref_counts["tri_tests"] + 10.0 * ref_counts["tri_past_det"] + 20.0 * ref_counts["tri_past_beta"]
                  + 17.0 * ref_counts["tri_past_gamma"])
        f += 150.0 * ref_counts["shades"]
    else:
        f += 40.0 * st["tri_tests"]
    f += 60.0 * st["shadow_rays"]
    return f


def cpu_sample_factor(desc, cam) -> int:
    """The reference's algorithm tests every instance's world box for every ray
    (utils.cpp:756-790), so a CPU pass over a 4K frame with thousands of instances takes
    minutes.  CPU legs of the bench are bounded: they render the same view with the image
    reduced by this integer factor in both directions, and scale by its square."""
    n_inst = int(desc.contents.n_instances)
    cost = float(cam.width) * float(cam.height) * (n_inst + 50.0)
    return max(1, int(math.ceil(math.sqrt(cost / 2.0e9))))


def reduced_camera(cam, f: int):
    from raytracer_amd import Camera
    out = Camera()
    C.memmove(C.byref(out), C.byref(cam), C.sizeof(Camera))
    out.width = max(1, cam.width // f)
    out.height = max(1, cam.height // f)
    return out


def oracle_frame(desc, cam, nthreads, want_hits=False):
    """One frame through the CPU restatement of the reference (oracle/librt_oracle.so, the
    checker): -> (work counts, rgb, hit records or None, ms), or (None,)*4 when it is not built."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "librt_oracle.so")
    if not os.path.exists(so):
        return None, None, None, None
    from raytracer_amd import _abi
    lib = C.CDLL(so)
    lib.rt_oracle_render.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.Camera), C.c_void_p, C.c_void_p,
                                     C.POINTER(OracleCounters), C.c_int]
    lib.rt_oracle_render.restype = C.c_double
    rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
    hits = np.empty(cam.height * cam.width, dtype=_abi.HIT_DTYPE) if want_hits else None
    ctr = OracleCounters()
    ms = lib.rt_oracle_render(desc, C.byref(cam), rgb.ctypes.data, hits.ctypes.data if want_hits else None,
                              C.byref(ctr), nthreads)
    if ms < 0:
        return None, None, None, None
    return {k: int(getattr(ctr, k)) for k, _ in OracleCounters._fields_}, rgb, hits, ms


def reference_work_counts(desc, cam, nthreads):
    counts, _, _, ms = oracle_frame(desc, cam, nthreads)
    return counts, ms


# ---------------------------------------------------------------------------------------------
# Workloads
# ---------------------------------------------------------------------------------------------
def scene_cache_dir():
    d = os.environ.get("RTX_BENCH_SCENES", os.path.join(tempfile.gettempdir(), f"rtx_bench_scenes_{os.getuid()}"))
    os.makedirs(d, exist_ok=True)
    return d


def make_workload(name: str, d: str, frame: int = 0, n_frames: int = SEQUENCE_FRAMES):
    """-> (scene json path, description string)"""
    from raytracer_amd import scenes
    if name == "dragon_metal":
        p = os.path.join(d, "dragon_metal.json")
        if not os.path.exists(p):
            scenes.scene_dragon_metal(d, faces=1_800_000, res=(1920, 1152), name="dragon_metal")
        return p, "dragon_metal stand-in: 1.8M-face smooth conductor PLY mesh + Lambertian instance + ground, 1920x1152, depth 6, 2 lights"
    if name == "dragon_metal_small":
        p = os.path.join(d, "dragon_metal_small.json")
        if not os.path.exists(p):
            scenes.scene_dragon_metal(d, faces=100_000, res=(480, 288), name="dragon_metal_small")
        return p, "dragon_metal stand-in, reduced: 100k faces, 480x288"
    if name == "marching_dragons":
        p = os.path.join(d, "marching_dragons.json")
        if not os.path.exists(p):
            scenes.scene_marching(d, faces=400_000, res=(1920, 1080), name="marching_dragons")
        return p, "marching_dragons stand-in: 400k-face base mesh + 8 transformed instances, 1920x1080"
    if name == "grass_desert":
        p = os.path.join(d, "grass_desert.json")
        if not os.path.exists(p):
            scenes.scene_grass(d, n_instances=10_000, res=(3840, 2160), name="grass_desert")
        return p, "grass_desert stand-in: 10k instanced blade clumps, 3840x2160"
    if name == "grass_small":
        p = os.path.join(d, "grass_small.json")
        if not os.path.exists(p):
            scenes.scene_grass(d, n_instances=400, res=(320, 180), name="grass_small")
        return p, "grass_desert stand-in, reduced: 400 instances, 320x180"
    if name == "metal_glass_plates":
        p = os.path.join(d, "metal_glass_plates.json")
        if not os.path.exists(p):
            scenes.scene_plates(d, res=(800, 800), name="metal_glass_plates")
        return p, "metal_glass_plates stand-in: plane, spheres, dielectric boxes of free triangles, 800x800"
    if name in ("camera_around_david", "camera_around_david_small"):
        small = name.endswith("_small")
        stem = "davids_small" if small else "davids_camera"
        p = os.path.join(d, f"{stem}_{frame:03d}.json")
        if not os.path.exists(p):
            scenes.scene_david_frame(d, frame, n_frames=n_frames, faces=4_000 if small else 100_000,
                                     res=(160, 160) if small else (800, 800), name=stem)
        return p, (f"camera_around_david stand-in{' (reduced)' if small else ''}: "
                   f"{'4k' if small else '100k'}-face mesh + mirror ground, {'160x160' if small else '800x800'}, "
                   f"{n_frames}-frame orbit")
    raise SystemExit(f"unknown workload {name}")


def is_sequence(name: str) -> bool:
    return name.startswith("camera_around_david")


def orbit_camera(cam, angle_deg: float):
    """Rotate the camera about the world y axis through the origin (the frames of the
    turntable that whole-frame mode times)."""
    from raytracer_amd import Camera
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    out = Camera()
    C.memmove(C.byref(out), C.byref(cam), C.sizeof(Camera))
    for name in ("position", "gaze", "up"):
        v = getattr(cam, name)
        x, y, z = v[0], v[1], v[2]
        getattr(out, name)[:] = [c * x + s * z, y, -s * x + c * z]
    return out


def turntable_frame(step: int, rank: int, world: int, frames: int) -> int:
    """Frame of a `frames`-frame turntable that rank `rank` of `world` renders in step `step`
    (negative steps are the warm-up): every rank goes once round in `frames` steps, rank r starting
    r/N of the way round, so all ranks do the same work whatever the frames cost."""
    return (step + (rank * frames) // world) % frames


def reduced_scene_file(scene_json: str, f: int) -> str:
    """Sibling of the scene file (mesh files are found relative to the scene's directory,
    scene.cpp:502) whose cameras render 1/f of the resolution in both directions."""
    with open(scene_json) as fh:
        j = json.load(fh)

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "ImageResolution" and isinstance(v, str):
                    w, h = (int(x) for x in v.split())
                    o[k] = f"{max(1, w // f)} {max(1, h // f)}"
                else:
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(j)
    out = scene_json[:-5] + f"_cpu_sample_{f}.json"
    with open(out, "w") as fh:
        json.dump(j, fh)
    return out


def read_ppm(path):
    """ASCII P3 file (scene.cpp:686-695) -> uint8 [h][w][3]."""
    import numpy as np
    with open(path, "rb") as f:
        tok = f.read().split()
    if tok[0] != b"P3":
        raise ValueError("not a P3 file")
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:4 + w * h * 3], dtype=np.int32).astype(np.uint8).reshape(h, w, 3)


def cpu_reference_baseline(scene_json: str, rays: int, sample_f: int = 1, runs: int = 3):
    """Time the unmodified reference renderer (oracle/_ref/raytracer_ref, built from the
    reference's own sources) on this host: `./raytracer <scene> -b`, multithreaded default
    = min(4, nproc) threads (raytracer.cpp:99-100).  Reads renderTimeMs from its results
    JSON (raytracer.cpp:269-282); median of `runs` runs.  sample_f > 1: bounded sample, see
    cpu_sample_factor.  -> (cpu_baseline object, the picture the reference wrote) or (None, None)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
    if not os.access(ref, os.X_OK):
        return None, None
    work = tempfile.mkdtemp(prefix="rtx_ref_")
    try:
        os.makedirs(os.path.join(work, "run"))
        out_dir = os.path.join(work, "my_outputs_hw2")
        os.makedirs(out_dir)
        if sample_f > 1:
            scene_json = reduced_scene_file(scene_json, sample_f)
        times, pre, walls = [], [], []
        image = None
        for _ in range(max(1, runs)):
            for fn in os.listdir(out_dir):
                os.remove(os.path.join(out_dir, fn))
            t0 = time.time()
            r = subprocess.run([ref, scene_json, "-b"], cwd=os.path.join(work, "run"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=900)
            walls.append(time.time() - t0)
            if r.returncode != 0:
                return None, None
            res = [f for f in os.listdir(out_dir) if f.endswith("_results.json")]
            if not res:
                return None, None
            with open(os.path.join(out_dir, res[0])) as f:
                j = json.load(f)
            times.append(float(j["renderTimeMs"]))
            pre.append(float(j["preprocessingTimeMs"]))
            if sum(walls) > 60.0:
                break
        ppm = [f for f in os.listdir(out_dir) if f.endswith(".ppm")]
        if ppm:
            try:
                image = read_ppm(os.path.join(out_dir, ppm[0]))
            except Exception:
                image = None
        times_sorted = sorted(times)
        render_ms = times_sorted[len(times_sorted) // 2] * sample_f * sample_f
        sample = "whole frame" if sample_f == 1 else \
            f"the same view at 1/{sample_f} of the resolution in both directions, time scaled by {sample_f * sample_f}"
        return {"value": rays / render_ms / 1e3, "unit": "Mrays/s", "cores": min(4, os.cpu_count() or 4),
                "kind": "reference", "ms_per_frame": render_ms, "runs_ms": [t * sample_f * sample_f for t in times],
                "preprocess_ms": sorted(pre)[len(pre) // 2], "host_cpus": os.cpu_count(),
                "sample": f"{sample}, median of {len(times)} runs of ./raytracer <scene> -b (the reference caps itself at "
                          f"4 threads); process wall {sum(walls):.1f} s in total"}, image
    except Exception:
        return None, None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def image_delta(a, b):
    import numpy as np
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"identical": float((d.reshape(-1, 3).max(axis=1) == 0).mean()), "max_delta": int(d.max()) if d.size else 0}


def hit_mismatches(a, b) -> int:
    import numpy as np
    bad = (a["kind"] != b["kind"]) | (a["container"] != b["container"]) | (a["face"] != b["face"]) | (a["src"] != b["src"])
    bad |= a["distance"].view(np.uint64) != b["distance"].view(np.uint64)
    return int(bad.sum())


# ---------------------------------------------------------------------------------------------
# One rank
# ---------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: start bench.py without WORLD_SIZE "
                             f"(it launches its own ranks) or under a launcher with {args.gpus} ranks")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: there is no CPU path behind the C-ABI")
        # Rehearsal on a one-GPU box (RTX_BENCH_REHEARSAL=1): every rank uses device 0 and the
        # collectives run over gloo on host tensors; everything else is the multi-GPU code path.
        self.rehearsal = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local_rank = 0
        elif self.local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} but {torch.cuda.device_count()} GPU(s) visible "
                             "(set RTX_BENCH_REHEARSAL=1 to rehearse several ranks on one GPU)")
        torch.cuda.set_device(self.local_rank)
        self.red_dev = "cpu" if self.rehearsal else "cuda"
        self.dist = None
        self.backend = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
            self.backend = dist.get_backend()
        self.host_threads = max(1, min(16, os.cpu_count() or 1) // max(self.world, 1))

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def note(self, msg):
        if self.rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def reduce(self, values, op="sum"):
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.red_dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def workload(self, name, frames=(0,)):
        """Scene files are generated once, by rank 0, into the shared cache directory."""
        d = scene_cache_dir()
        if self.rank == 0:
            for f in frames:
                make_workload(name, d, f)
        self.barrier()
        return [make_workload(name, d, f) for f in frames]

    def pinned(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8).pin_memory().numpy()

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def frame_hash(frame) -> int:
    """63-bit hash of a picture's bytes (blake2b), for comparing pictures between ranks."""
    import hashlib
    import numpy as np
    h = hashlib.blake2b(np.ascontiguousarray(frame).tobytes(), digest_size=8).digest()
    return int.from_bytes(h, "little") & 0x7FFFFFFFFFFFFFFF


def multi_rank_check(R, frame, ms_per_step):
    """What lets a reader check an N > 1 line without the hardware: the backend of the process
    group, the number of ranks the collective saw (an all-reduce of ones; the run fails if it is
    not N), every rank's own time per step, and whether `frame` -- a picture every rank rendered
    from the same camera -- is the same bytes on every rank (an all-gather of 63-bit hashes)."""
    torch = R.torch
    seen = int(round(R.reduce([1.0])[0]))
    if seen != R.world:
        raise SystemExit(f"rank {R.rank}: the collective saw {seen} ranks, expected {R.world}")
    mine = torch.tensor([frame_hash(frame)], dtype=torch.int64, device=R.red_dev)
    per = torch.tensor([float(ms_per_step)], dtype=torch.float64, device=R.red_dev)
    if R.dist is not None:
        hashes = [torch.zeros_like(mine) for _ in range(R.world)]
        times = [torch.zeros_like(per) for _ in range(R.world)]
        R.dist.all_gather(hashes, mine)
        R.dist.all_gather(times, per)
    else:
        hashes, times = [mine], [per]
    hashes = [int(h.item()) for h in hashes]
    times = [float(t.item()) for t in times]
    return {"backend": R.backend, "ranks_seen": seen, "ms_per_step_per_rank": times,
            "ms_per_step_min": min(times), "ms_per_step_max": max(times),
            "frames_identical_across_ranks": all(h == hashes[0] for h in hashes),
            "frame_hashes": [f"{h:016x}" for h in hashes]}


def timed(R: Rank, step, steps, warmup):
    """warmup untimed steps, then exactly `steps` steps between barrier + synchronize on both
    sides -> (elapsed seconds, max over ranks; list of the steps' rtx_stats; this rank's own seconds)."""
    for i in range(warmup):
        step(i - warmup)
    R.barrier()
    t0 = time.perf_counter()
    stats = [step(i) for i in range(steps)]
    mine = time.perf_counter() - t0            # (this rank's own K steps, before it waits for the others)
    R.barrier()
    elapsed = time.perf_counter() - t0
    return R.reduce([elapsed], "max")[0], stats, mine


def pmc_summary(workload):
    """Hardware-counter figures of the TIMED trace-kernel variant from the committed profile of this
    workload (profiles/<round>_<workload>_pmc.json, written by tools/pmc_to_json.py from separate
    rocprofv3 --pmc passes over this same bench; per kernel variant, so that the counters-on variant of
    the untimed counting pass is not mixed in), or None.  The object names the build the profile was
    taken from: these figures are copied from the file, not measured in this run."""
    p = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{workload}_pmc.json")
    try:
        with open(p) as f:
            j = json.load(f)
        v = j["variants"][j["timed_variant"]]
        return {"source": f"profiles/{PROFILE_ROUND}_{workload}_pmc.json (build {j.get('build', '?')}; copied from that file, "
                          "not measured in this run)", "kernel": j["timed_variant"], **v}
    except Exception:
        return None


def measure_static(R: Rank, args, name, mode, steps, warmup, with_cpu, pipelined=False):
    """Workloads with one scene file.  mode frames: every rank renders each frame of a turntable
    of `steps` frames once, rank r starting r/N of the way round.
    mode tiles: all ranks render interleaved tiles of ONE frame, gathered on rank 0."""
    import numpy as np
    import raytracer_amd as rt
    from raytracer_amd import distributed as D
    (path, descr), = R.workload(name)
    R.note(f"workload {name} ({mode}): loading scene")
    t0 = time.time()
    hs = rt.HostScene(path)
    t_load = time.time() - t0
    t0 = time.time()
    ds = rt.DeviceScene(hs.desc, R.local_rank)
    t_upload = time.time() - t0
    cam0, _ = hs.camera(0)
    tw, th = (int(x) for x in args.tile.split("x"))
    world, rank = R.world, R.rank
    cam = cam0
    # mode frames: the timed steps are a turntable of `steps` frames around the scene (frame 0 = the
    # scene file's own camera); every rank renders every frame of it exactly once, rank r starting
    # r/N of the way round.  The views differ in cost by up to 40 %, so a rank that kept one view
    # would make the slowest view the measure of every N > 1; this way each GPU does the same work
    # at every N, which is what weak scaling compares.
    turntable = [orbit_camera(cam0, 0.0 if args.scene_view else 360.0 * j / steps) for j in range(steps)] if mode == "frames" else None

    # ---- untimed counting pass: rays and traversal work of this rank's step ----------------
    R.note("device counting pass")
    ds.set_counters(True)
    if mode == "frames":
        _, _, cst = ds.render(cam)
    else:
        _, cst = ds.render_tiles(cam, tw, th, rank, world)
    ds.set_counters(False)
    rays_rank = int(cst["rays"])
    sample_f = cpu_sample_factor(hs.desc, cam)
    ref_counts = o_rgb = o_hits = None
    ocam = cam
    if rank == 0:
        if sample_f > 1:
            # the view the bounded CPU legs render: the camera of the scene file with its resolution
            # reduced (the reference derives the near plane from the file's own aspect ratio), so that
            # oracle, device and the reference's PPM are pictures of exactly the same camera
            small = rt.HostScene(reduced_scene_file(path, sample_f))
            ocam, _ = small.camera(0)
            small.close()
        R.note(f"reference work counts on the host (sample factor {sample_f})")
        ref_counts, o_rgb, o_hits, _ = oracle_frame(hs.desc, ocam, R.host_threads, want_hits=True)
    counters_agree = None
    if ref_counts is not None and sample_f > 1:
        # bounded sample: the same view at 1/f of the resolution in both directions, counts scaled
        # to the full frame by the device's own ray count
        scale = rays_rank / max(ref_counts["rays"], 1)
        ref_scaled = {k: int(round(v * scale)) for k, v in ref_counts.items()}
        ref_b = reference_bytes(ref_scaled)
        alg_source = (f"reference algorithm's work counts (CPU restatement on the same view at 1/{sample_f} resolution, "
                      "scaled to the frame's ray count) x reference record sizes")
    elif ref_counts is not None:
        # (world-box tests are not compared: the device's TLAS rejects most instances before
        # their boxes are tested, the reference tests every instance's box, utils.cpp:756-766)
        counters_agree = ref_counts["rays"] == cst["rays"] and ref_counts["tri_tests"] == cst["tri_tests"]
        ref_scaled = ref_counts
        ref_b = reference_bytes(ref_counts)
        alg_source = "reference algorithm's work counts (CPU restatement, untimed pass) x reference record sizes"
    else:
        ref_scaled, ref_b, alg_source = None, None, None
    flops_counted = frame_flops(cst, ref_scaled)

    # ---- parity of this very frame (rank 0): device (default mode) against the checker -------
    parity = None
    if o_rgb is not None:
        d_rgb, d_hits, _ = ds.render(ocam, want_hits=True)
        parity = {"view": f"{ocam.width}x{ocam.height}" + ("" if sample_f == 1 else f" (the frame's view at 1/{sample_f})"),
                  "vs_oracle": {**image_delta(d_rgb, o_rgb), "hit_mismatches": hit_mismatches(d_hits, o_hits),
                                "pixels": int(ocam.width * ocam.height)}}
        parity_rgb = d_rgb.copy()

    # the caller-owned image buffer of the C-ABI: page-locked and reused from frame to frame
    frame = R.pinned(cam.height * cam.width * 3).reshape(cam.height, cam.width, 3)
    if args.consistent:
        ds.set_consistent(True)
    if mode == "frames":
        def step(i):
            return ds.render(turntable[turntable_frame(i, rank, world, steps)], out=frame)[2]
    else:
        gatherer = D.TileGatherer(cam.width, cam.height, tw, th, rank, world, device=R.red_dev,
                                  unpack=rt.unpack_tiles, frame=frame, assemble=rt.assemble_tiles)
        last = {}

        def render_tiles(first, stride, out):
            _, st = ds.render_tiles(cam, tw, th, first, stride, out_ptr=out.data_ptr(), out_is_device=out.is_cuda)
            last["st"] = st

        def step(i):
            gatherer.run(render_tiles)
            return last["st"]
    R.note("timed steps")
    elapsed, stats, elapsed_mine = timed(R, step, steps, warmup)
    # Several frames in flight per GPU (throughput of a stream of frames rather than the latency of
    # one): frame contexts of the ONE device scene (rtx_frame_create), each frame submitted without
    # waiting (rtx_render_async) and collected when its context's turn comes round again; one frame's
    # thin last generations overlap the next ones' full generations.  Every frame of the turntable is
    # rendered k_fl times in all.
    piped = None
    if pipelined and mode == "frames":
        k_fl = args.inflight if args.inflight > 1 else 4          # (tools/inflight_sweep.py: 2, 4 and 6 in flight beat 3 on dragon_metal)
        ds.set_generation_timing(False)      # (as the command line's pipeline runs: no event pairs round the generations)
        slots = [(ds.frame(), R.pinned(cam.height * cam.width * 3).reshape(cam.height, cam.width, 3)) for _ in range(k_fl)]
        order = [turntable_frame(i, rank, world, steps) for i in range(steps)]

        def stream(n):
            q = []
            for i in range(n):
                fr, buf = slots[i % k_fl]
                if len(q) == k_fl:
                    q.pop(0).wait()
                fr.submit(turntable[order[i % steps]], out=buf)
                q.append(fr)
            for fr in q:
                fr.wait()
        stream(k_fl)
        R.barrier()
        t0 = time.perf_counter()
        stream(k_fl * steps)
        R.barrier()
        el2 = R.reduce([time.perf_counter() - t0], "max")[0]
        # the last picture of every context against the blocking call's
        same = True
        for j, (fr, buf) in enumerate(slots):
            i = k_fl * steps - k_fl + j
            same = same and bool(np.array_equal(buf, ds.render(turntable[order[i % steps]])[0]))
        for fr, _ in slots:
            fr.close()
        ds.set_generation_timing(True)
        piped = {"frames_in_flight_per_gpu": k_fl, "device_scenes_per_gpu": 1, "frames": k_fl * steps * world, "wall_s": el2,
                 "ms_per_frame_per_gpu": el2 / (k_fl * steps) * 1e3, "identical_to_blocking_render": same}
    # The eight-wide float-box records (rtx_scene_set_wide_blas; off by default): the same turntable once
    # more, reported beside the headline -- the same bytes from fewer records.
    wide = None
    if pipelined and mode == "frames":
        # (its own device scene, created with the records: the timed one above holds none)
        first_cam = turntable[turntable_frame(0, rank, world, steps)]
        narrow_first = ds.render(first_cam)[0].copy()
        timed_ds, ds = ds, rt.DeviceScene(hs.desc, R.local_rank, wide_blas=True)
        ds.set_wide_blas(True)
        w_first = ds.render(first_cam)
        if w_first[2]["wide_blas"]:
            same = bool(np.array_equal(w_first[0], narrow_first))
            el_w, st_w, _ = timed(R, step, steps, 1)
            wide = {"ms_per_frame": el_w / steps * 1e3, "record_fetches_per_frame": sum(s_["record_fetches"] for s_ in st_w) / steps,
                    "trace_ms_per_frame": sum(s_["trace_ms"] for s_ in st_w) / steps,
                    "redone_frames": int(sum(s_["wide_blas_redone"] for s_ in st_w)),
                    "identical_to_default_path": same,
                    "what": "rtx_scene_set_wide_blas: three levels of the reference's BVH per 256-byte record, float boxes rounded "
                            "outward, one stack entry per record, every job's result checked against the reference's exact leaf box",
                    "record_bytes": ds.record_bytes(), "record_bytes_default": timed_ds.record_bytes()}
            # The non-parity mode (rtx_scene_set_consistent: bounds carried into object space, any-hit shadow
            # rays) over both record kinds -- its own figures, never the headline's.
            if not args.consistent:
                ds.set_consistent(True)
                cw_first = ds.render(first_cam)[0].copy()
                el_cw, _, _ = timed(R, step, steps, 1)
                ds.set_consistent(False)
                wide_ds, ds = ds, timed_ds
                ds.set_consistent(True)
                c_first = ds.render(first_cam)[0].copy()
                el_c, _, _ = timed(R, step, steps, 1)
                ds.set_consistent(False)
                ds = wide_ds
                wide["consistent_mode"] = {"ms_per_frame": el_c / steps * 1e3, "ms_per_frame_wide_blas": el_cw / steps * 1e3,
                                           "wide_blas_identical_to_four_wide": bool(np.array_equal(c_first, cw_first)),
                                           "note": "not the reference's picture where instances are scaled or buried; checked against its own "
                                                   "CPU checker in the tests"}
        ds.close()
        ds = timed_ds
    # mode tiles: the assembled frame against the same frame rendered whole on rank 0
    tiles_identical = None
    if mode == "tiles" and rank == 0:
        whole = ds.render(cam)[0]
        tiles_identical = bool(np.array_equal(whole, frame))
    kernel_ms = sum(s["kernel_ms"] for s in stats)
    trace_ms = sum(s["trace_ms"] for s in stats)
    trace_launches = sum(s["trace_launches"] for s in stats)
    # rays of the timed frames (rtx_stats counts every frame's rays without device counters)
    rays_timed_rank = sum(int(s_["rays"]) for s_ in stats)
    rays_step = R.reduce([rays_timed_rank])[0] / steps
    rays_frame = rays_timed_rank / steps              # this rank's mean frame
    # the scene file's own view alone (the frame the CPU reference is timed on and parity is checked on)
    view0_ms = None
    if mode == "tiles":
        view0_ms = elapsed / steps * 1e3          # (the tiled frame is the scene file's own view)
    if mode == "frames" and with_cpu:
        ds.render(cam0, out=frame)
        t0 = time.perf_counter()
        for _ in range(5):
            ds.render(cam0, out=frame)
        view0_ms = (time.perf_counter() - t0) / 5 * 1e3
    multi = None
    if world > 1 or args.multi_check:
        # every rank renders one and the same view once, untimed, and the pictures' hashes are compared
        # (tiles mode: the view at a quarter of the resolution, rendered whole)
        ccam = cam0 if mode == "frames" else reduced_camera(cam0, 4)
        multi = multi_rank_check(R, ds.render(ccam)[0], elapsed_mine / steps * 1e3)
    out = None
    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        # What the timed frames fetched (the default mode may take another path than the counters-on
        # frame: in scenes with many instances it searches them nearest first instead of replaying
        # them in index order); the arithmetic model, built on the counters, is scaled by the same ratio.
        rec_timed = sum(s_["record_fetches"] for s_ in stats) / max(len(stats), 1)
        rec_counted = max(int(cst["record_fetches"]), 1)
        early_timed = sum(s_["shadow_rays_settled_early"] for s_ in stats) / max(len(stats), 1)
        dev_b = device_bytes(cst, rec_timed, early_timed)
        flops = flops_counted * (rec_timed / rec_counted)
        # Dominant kernel: k_trace, the closest-hit traversal, launched once per ray generation (two
        # variants per generation, one of which returns at once).  Every generation's launches are
        # bracketed by HIP events on the library's own stream (rtx_stats.trace_ms / trace_launches).
        launches_per_frame = trace_launches / steps
        avg_launch_ms = trace_ms / max(trace_launches, 1)
        # algorithmic bytes: the reference's algorithm in its record sizes -- unless the device's own
        # records for the same frame are fewer bytes (it replaces the reference's linear instance loop
        # by a TLAS): then those, so that the fraction never counts bytes nobody has to move
        basis = "reference" if (ref_b is not None and ref_b <= dev_b) else "device"
        alg_b = ref_b if basis == "reference" else dev_b
        bytes_per_launch = alg_b / max(launches_per_frame, 1)
        t_launch = avg_launch_ms * 1e-3
        achieved = bytes_per_launch / t_launch / 1e9 if t_launch > 0 else 0.0
        t_trace_frame = trace_ms / steps * 1e-3
        pmc = pmc_summary(name)
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        tflops = flops / t_trace_frame / 1e12 if t_trace_frame > 0 else 0.0
        # The stated bound: a scene whose records (nodes, triangles, object records) fit the L2s is
        # served out of them, whatever the algorithmic byte count says -- HBM sees next to nothing of
        # it (the counter traffic beside it shows that); a larger scene is priced against HBM.
        working_set = ds.record_bytes()
        l2_bound = 0 < working_set <= L2_BYTES
        l2_rate = dev_b / t_trace_frame / 1e9 if t_trace_frame > 0 else 0.0
        hbm = {"algorithmic_achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_algorithmic": achieved / HBM_PEAK_GBS,
               "measured_bytes_per_launch": traffic,
               "measured_achieved": (traffic / t_launch / 1e9) if (traffic and t_launch > 0) else None,
               "frac_measured": (traffic / t_launch / 1e9 / HBM_PEAK_GBS) if (traffic and t_launch > 0) else None,
               "measured_source": pmc["source"] if pmc else None,
               "note": "algorithmic bytes / kernel time against the HBM peak, and beside it the HBM-side bytes the counters "
                       "saw (2 x FETCH_SIZE + WRITE_SIZE per launch of the timed variant) over this run's launch time"}
        roofline = {
            "bound": "l2" if l2_bound else "hbm",
            "achieved": l2_rate if l2_bound else achieved, "peak": L2_PEAK_GBS if l2_bound else HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (l2_rate / L2_PEAK_GBS) if l2_bound else achieved / HBM_PEAK_GBS, "traffic": traffic,
            "record_working_set_bytes": working_set,
            "hbm": hbm,
            "kernel": "k_trace (closest-hit traversal; all variants of a generation)",
            "launches_per_frame": launches_per_frame, "avg_launch_ms": avg_launch_ms,
            "trace_ms_per_frame": trace_ms / steps, "all_kernels_ms_per_frame": kernel_ms / steps,
            "algorithmic_basis": basis,
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_frame": alg_b,
            "algorithmic_bytes_per_ray": alg_b / max(rays_frame if basis == "device" else rays_rank, 1),
            "reference_bytes_per_frame": ref_b, "reference_bytes_source": alg_source,
            "device_bytes_per_frame": dev_b,
            "device_bytes_source": "records the trace kernel fetched in the timed frames (rtx_stats.record_fetches, counted in "
                                   "every mode) x 256 B, + 96 B per queued ray (shadow rays settled by the shading kernel are not queued) + 128 B per hit record",
            "record_fetches_per_frame": rec_timed,
            "shadow_rays_settled_early_per_frame": early_timed,
            "record_fetches_per_frame_counters_on": rec_counted,
            "instance_path": {0: "candidates replayed in index order", 1: "nearest-first search with verification (identical "
                              "results; rtx_kernels.h, search_service)", 2: "consistent mode's nearest-first walk"}.get(
                                  int(stats[-1].get("instance_path", 0)), "?"),
            "l2": {"achieved": l2_rate, "peak": L2_PEAK_GBS,
                   "unit": "GB/s", "frac": l2_rate / L2_PEAK_GBS,
                   "gather_peak": L2_GATHER_GBS, "frac_of_gather_peak": l2_rate / L2_GATHER_GBS,
                   "note": "device record bytes of the frame / trace time: the rate at which records leave the caches "
                           "(most of them L2 or Infinity Cache hits), against the aggregate L2 bandwidth and against the "
                           "rate at which whole records can be gathered into LDS out of L2 (measured, 16.8-18.8 TB/s)"},
            "valu": {"flops_per_frame": flops, "achieved": tflops, "unit": "TFLOP/s",
                     "fp64_peak": FP64_VALU_PEAK_TF, "frac_fp64": tflops / FP64_VALU_PEAK_TF,
                     "fp32_peak": FP32_VALU_PEAK_TF, "frac_fp32": tflops / FP32_VALU_PEAK_TF,
                     "scaled_by_record_fetches": rec_timed / rec_counted,
                     "model": "FP64 operations of the device's own arithmetic: 100 per four-wide record visit (4 slab tests "
                              "of 25), triangle tests by the stage they reach (19 / +10 / +20 / +17), 25 per world-box "
                              "test, 53 per ray taken to object space, 150 per shaded hit, 60 per shadow ray; counts from "
                              "the device counters and the CPU restatement's stage counters"},
            "counters_agree": counters_agree,
            "reference_work_counts": ref_scaled,
            "device_record_visits": int(cst["node_visits"]), "tri_tests": int(cst["tri_tests"]),
            "device_world_box_tests": int(cst["world_box_tests"]),
            "device_object_space_traversals": int(cst["object_space_traversals"]),
            "device_tlas_node_visits": int(cst["tlas_node_visits"]),
            "lanes_at_nodes_per_wave_step": cst["wave_inner_lanes"] / max(cst["wave_inner_steps"], 1),
            "lanes_at_leaves_per_wave_step": cst["wave_leaf_lanes"] / max(cst["wave_leaf_steps"], 1),
            "occupancy_waves_per_simd": 2,
            "pmc": pmc,
        }
        out = {
            "metric": "Mrays/s (ms/frame in ms_per_step)", "value": rays_step * steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if mode == "frames" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name, "description": descr, "mode": mode,
                       "sharding": ("whole frames: every step renders the scene file's own view (--scene-view)" if args.scene_view and mode == "frames" else
                                    f"whole frames of a {steps}-frame turntable around the scene (frame 0 = the scene file's camera): "
                                    "every rank renders each frame once, rank r starting r/N of the way round; no collective"
                                    if mode == "frames" else
                                    f"interleaved {tw}x{th} tiles of one frame, one gather to rank 0 per frame"),
                       "assembled_frame_identical_to_whole_frame": tiles_identical,
                       "resolution": [cam.width, cam.height], "rays_per_step": rays_step,
                       "rays_per_pixel": rays_frame / (cam.width * cam.height) if mode == "frames" else None,
                       "rays_scene_view": rays_rank if mode == "frames" else None,
                       "frames_per_step": world if mode == "frames" else 1,
                       "host_load_s": t_load, "scene_upload_s": t_upload,
                       "image_buffer": "host, page-locked, reused (download inside the timed step)",
                       "traversal": ("consistent mode: not the reference's results where instances are scaled; rays and "
                                     "algorithmic bytes are those of the same frame in the default mode")
                                    if args.consistent else "reference-exact (default)"},
            "roofline": roofline,
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if wide is not None:
            wide["record_fetches_vs_default"] = wide["record_fetches_per_frame"] / max(rec_timed, 1)
            wide["ms_per_frame_vs_default"] = wide["ms_per_frame"] / ms_per_step
            out["wide_blas"] = wide
        if piped is not None:
            piped["Mrays_per_s"] = rays_step * piped["frames_in_flight_per_gpu"] * steps / piped["wall_s"] / 1e6
            out["frames_in_flight"] = piped
        if with_cpu:
            R.note("reference CPU renderer")
            cb, ref_img = cpu_reference_baseline(path, rays_rank if mode == "frames" else int(rays_step), sample_f)
            if cb is not None:
                out["cpu_baseline"] = cb
                # like for like: the reference's frame is the scene file's own view
                out["config"]["scene_view_ms_per_frame"] = view0_ms
                out["config"]["speedup_vs_cpu_reference_ms_per_frame"] = cb["ms_per_frame"] / (view0_ms or ms_per_step)
                # (no published figure exists for this metric on this hardware: the baseline is the reference
                # itself, timed in this run on this box's host cores, frame time against frame time)
                out["vs_baseline"] = cb["ms_per_frame"] / (view0_ms or ms_per_step)
                out["config"]["vs_baseline_is"] = "cpu_baseline.ms_per_frame / scene_view_ms_per_frame (same view, same run, same box)"
                if parity is not None and ref_img is not None and ref_img.shape == parity_rgb.shape:
                    parity["vs_reference_ppm"] = image_delta(parity_rgb, ref_img)
        if parity is not None:
            out["parity"] = parity
    ds.close()
    hs.close()
    return out


def measure_sequence(R: Rank, args, name, steps, warmup, whole, inflight=1, parity_frames=0):
    """camera_around_david: rank r renders frames r, r + N, ... of the 360-frame sequence through
    the animation pipeline (geometry loaded and uploaded once per rank; per frame the scene
    file is re-read for camera and lights only).  whole: every frame of the sequence once
    (steps is ignored); otherwise `steps` frames per rank after `warmup` earlier ones.
    inflight > 1: that many frames of a rank are in flight at once on the rank's ONE device scene,
    each in a frame context of its own (rtx_frame_create; submitted with rtx_render_async, collected
    when the context's turn comes round again): while one frame's last, thin ray generations drain --
    a few long rays in an otherwise idle GPU -- the next frames' full generations run beside them.
    parity_frames > 0 (rank 0): that many frames spread over the sequence are rendered once more,
    untimed, and compared with the CPU restatement of the reference (hit records and picture)."""
    import numpy as np
    import raytracer_amd as rt
    from raytracer_amd import distributed as D
    world, rank = R.world, R.rank
    n_seq = SEQUENCE_FRAMES
    if whole:
        mine = D.shard_frames(n_seq, rank, world)
        warm = mine[:1]
    else:
        mine = [(rank + i * world) % n_seq for i in range(steps)]
        warm = [(rank + (i - warmup) * world) % n_seq for i in range(warmup)]
    needed = sorted(set(range(n_seq)) if whole else set(f for r in range(world) for i in range(-warmup, steps)
                                                        for f in [(r + i * world) % n_seq]))
    check_ranks = world > 1 or getattr(args, "multi_check", False)
    if check_ranks and needed[0] != 0:
        needed = [0] + needed           # (the frame every rank renders once for the self-check below)
    files = dict(zip(needed, R.workload(name, needed)))
    descr = files[needed[0]][1]
    R.note(f"sequence {name}: loading the geometry once per rank")
    t0 = time.time()
    k_fl = max(1, inflight)
    hs = rt.HostScene(files[mine[0]][0])
    ds = rt.DeviceScene(hs.desc, R.local_rank)
    cam, _ = hs.camera(0)
    if k_fl > 1:
        # as the command line's animation pipeline runs: no event pairs round the generations
        # (rtx_stats.trace_ms stays 0; they cost an 800x800 frame 3-4 %)
        ds.set_generation_timing(False)
    slots = [(ds.frame(), R.pinned(cam.height * cam.width * 3).reshape(cam.height, cam.width, 3)) for _ in range(k_fl)]
    t_setup = time.time() - t0
    lights_now = [None]

    def next_frame(f):
        """The host side of a frame: its file re-read for camera and lights (geometry kept)."""
        if not hs.load_next(files[f][0]):
            raise SystemExit(f"frame {f}: the geometry was not kept")
        lights = hs.lights()
        key = b"".join(bytes(l) for l in lights)
        if key != lights_now[0]:                       # (a new version of the lights only when they changed)
            ds.update_lights(lights)
            lights_now[0] = key
        return hs.camera(0)[0]

    def render_all(frames):
        """-> list of rtx_stats, in frame order"""
        stats, q = [], []
        for i, f in enumerate(frames):
            fr, buf = slots[i % k_fl]
            if len(q) == k_fl:
                stats.append(q.pop(0).wait()[2])
            fr.submit(next_frame(f), out=buf)
            q.append(fr)
        for fr in q:
            stats.append(fr.wait()[2])
        return stats

    # rays need no device counters (rtx_stats fills them from the queue fill levels), so the
    # timed frames themselves give the count
    R.note(f"sequence: timed frames ({k_fl} in flight)")
    render_all(warm * k_fl)
    R.barrier()
    t0 = time.perf_counter()
    stats = render_all(mine)
    own_s = time.perf_counter() - t0               # (this rank's own frames, before it waits for the others)
    R.barrier()
    elapsed = R.reduce([time.perf_counter() - t0], "max")[0]
    rays_mine = sum(int(s["rays"]) for s in stats)
    rays_all, frames_all = R.reduce([rays_mine, len(mine)])
    parity = None
    if parity_frames > 0 and rank == 0:
        R.note(f"sequence: {parity_frames} frames against the CPU restatement")
        checked, worst_same, worst_delta, bad_hits = [], 1.0, 0, 0
        for k in range(parity_frames):
            f = needed[(k * len(needed)) // parity_frames]
            c = next_frame(f)
            rgb, hits, _ = ds.render(c, want_hits=True)
            _, o_rgb, o_hits, _ = oracle_frame(hs.desc, c, R.host_threads, want_hits=True)
            if o_rgb is None:
                checked = None
                break
            d = image_delta(rgb, o_rgb)
            worst_same, worst_delta = min(worst_same, d["identical"]), max(worst_delta, d["max_delta"])
            bad_hits += hit_mismatches(hits, o_hits)
            checked.append(f)
        if checked:
            parity = {"frames_checked": checked, "vs_oracle": {"identical_min": worst_same, "max_delta": worst_delta,
                                                                "hit_mismatches": bad_hits,
                                                                "pixels_per_frame": int(cam.width * cam.height)}}
    # N > 1: the run checks itself (every rank seen, every rank's time, the same bytes for the same frame)
    multi = None
    if check_ranks:
        multi = multi_rank_check(R, ds.render(next_frame(0))[0], own_s / max(len(mine), 1) * 1e3)
    out = None
    if rank == 0:
        per_rank = max(len(D.shard_frames(n_seq, r, world)) for r in range(world)) if whole else steps
        out = {"workload": name, "description": descr,
               "sharding": "whole frames: frame f of the sequence on rank f mod N, no collective; geometry parsed, built "
                           "and uploaded once per rank, camera and lights re-read per frame",
               "frames": int(frames_all), "frames_per_rank": per_rank, "frames_in_flight_per_gpu": k_fl,
               "device_scenes_per_gpu": 1,
               "generation_timing": inflight <= 1, "wall_s": elapsed,
               "frames_per_s": frames_all / elapsed, "ms_per_frame_per_gpu": elapsed / per_rank * 1e3,
               "Mrays_per_s": rays_all / elapsed / 1e6, "rays": int(rays_all), "resolution": [cam.width, cam.height],
               "setup_s_per_rank": t_setup, "kernel_ms_per_frame": sum(s["kernel_ms"] for s in stats) / max(len(stats), 1),
               "trace_ms_per_frame": sum(s["trace_ms"] for s in stats) / max(len(stats), 1),
               "trace_launches_per_frame": sum(s["trace_launches"] for s in stats) / max(len(stats), 1),
               "n_gpus": world}
        if parity is not None:
            out["parity_sampled"] = parity
        if multi is not None:
            out["multi_gpu"] = multi
    for fr, _ in slots:
        fr.close()
    ds.close()
    hs.close()
    return out


def run_rank(args) -> int:
    R = Rank(args)
    world = R.world
    extras = (not args.no_extras and args.workload == "dragon_metal" and args.mode == "frames" and not args.consistent)
    if is_sequence(args.workload):
        seq = measure_sequence(R, args, args.workload, args.steps, args.warmup, whole=False, inflight=args.inflight,
                               parity_frames=0 if args.no_cpu_baseline else 6)
        out = None
        if R.rank == 0:
            out = {"metric": "Mrays/s (ms/frame in ms_per_step)", "value": seq["Mrays_per_s"], "unit": "Mrays/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": seq["wall_s"] / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": args.workload, "description": seq["description"], "mode": "frames",
                              "sharding": seq["sharding"], "frames_per_step": world, "frames_per_s": seq["frames_per_s"],
                              "rays_per_step": seq["rays"] / args.steps, "resolution": seq["resolution"]},
                   "sequence": seq}
            if seq.get("multi_gpu") is not None:
                out["multi_gpu"] = seq["multi_gpu"]
    else:
        out = measure_static(R, args, args.workload, args.mode, args.steps, args.warmup,
                             with_cpu=(world == 1 and not args.no_cpu_baseline), pipelined=extras or args.pipelined)
    if extras:
        # the two other shardings of the path, measured after the main one and reported beside it
        seq = measure_sequence(R, args, "camera_around_david", 0, 0, whole=True, parity_frames=12)
        seq2 = measure_sequence(R, args, "camera_around_david", 0, 0, whole=True, inflight=3)
        # the metric's second scene, with its own roofline, CPU baseline and parity (N = 1: the reference on a
        # 1/7 view, ~15 s of wall)
        tiles = measure_static(R, args, "grass_desert", "tiles", 5, 1, with_cpu=(world == 1 and not args.no_cpu_baseline))
        if R.rank == 0:
            out["sequence"] = seq
            out["sequence"]["frames_in_flight"] = {k: seq2[k] for k in ("frames_in_flight_per_gpu", "device_scenes_per_gpu", "generation_timing",
                                                                         "wall_s", "frames_per_s", "ms_per_frame_per_gpu", "Mrays_per_s")}
            out["tiled_frame"] = {k: tiles[k] for k in ("value", "unit", "ms_per_step", "n_gpus", "steps", "scaling")}
            out["tiled_frame"].update({"workload": "grass_desert", "sharding": tiles["config"]["sharding"],
                                       "resolution": tiles["config"]["resolution"],
                                       "assembled_frame_identical_to_whole_frame": tiles["config"]["assembled_frame_identical_to_whole_frame"],
                                       "rays_per_frame": tiles["config"]["rays_per_step"],
                                       "trace_ms_per_frame_rank0": tiles["roofline"]["trace_ms_per_frame"],
                                       "device_bytes_per_frame_rank0": tiles["roofline"]["device_bytes_per_frame"],
                                       "roofline": tiles["roofline"]})
            for k in ("cpu_baseline", "parity", "multi_gpu", "vs_baseline"):
                if tiles.get(k) is not None:
                    out["tiled_frame"][k] = tiles[k]
            if tiles["config"].get("speedup_vs_cpu_reference_ms_per_frame") is not None:
                out["tiled_frame"]["speedup_vs_cpu_reference_ms_per_frame"] = tiles["config"]["speedup_vs_cpu_reference_ms_per_frame"]
    if R.rank == 0:
        print(json.dumps(out), flush=True)
    R.close()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dragon_metal")
    ap.add_argument("--mode", choices=["frames", "tiles"], default="frames")
    ap.add_argument("--tile", default="64x8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene-view", action="store_true",
                    help="mode frames: every step renders the scene file's own camera instead of a turntable frame "
                         "(comparisons between builds; the frame of the parity check and of the CPU baseline)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sequence and tiled-frame measurements that follow the default workload")
    ap.add_argument("--consistent", action="store_true",
                    help="time the library's consistent mode (NOT the reference's results for scaled instances; "
                         "never the headline number).  Rays and algorithmic bytes stay those of the same frame in "
                         "the default, reference-exact mode")
    ap.add_argument("--inflight", type=int, default=1,
                    help="frames in flight per GPU: for camera_around_david as the main workload (default 1), and for --pipelined")
    ap.add_argument("--pipelined", action="store_true",
                    help="also measure the workload with several frames in flight per GPU (--inflight, at least 2; reported as frames_in_flight)")
    ap.add_argument("--multi-check", action="store_true",
                    help="report the multi_gpu self-check object at N = 1 too")
    ap.add_argument("--selftest-ranks", action="store_true",
                    help="check the rank launch path only: rendezvous over gloo, no GPU work")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-differ-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be at least 1 and --warmup at least 0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)
    if args.selftest_ranks:
        return selftest_rank(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
