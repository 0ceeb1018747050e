#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one frame of the workload through rtx_render.

Metric: Mrays/s and ms/frame.  A ray is one closest-hit query the reference
semantics require (one intersect() call, utils.cpp:671, not counting calls whose result
computePixelColor discards); the per-frame ray count R is taken from the device's own
counters in an untimed pass and is, by construction and by test, the count the CPU
restatement produces.

N = 1: the dragon_metal stand-in (one 1.8 M-face smooth conductor mesh + a Lambertian
instance of it + ground, 1920x1152, depth 6) unless --workload says otherwise.
N > 1: whole frames are sharded -- rank r renders orbit frame r of the same scene every
step, no collective on the data path (weak scaling); `--mode tiles` instead splits ONE
frame into interleaved tiles and gathers them on rank 0 over RCCL (strong scaling).

Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# Algorithmic traffic of one frame, in the reference's own in-memory record sizes: what the
# reference's traversal has to read per unit of work, whatever layout the device uses.
#   72 B  one BVH node whose box is tested           (sizeof(BVHNode), bvh.h:50-61)
#   88 B  one triangle test                          (4 B faceIndices + 12 B face + 3 x 24 B vertices)
#   48 B  one world-space AABB test                  (AABB, bvh.h:11-16)
#  128 B  one ray taken to object space              (one Matrix4x4, scene.h:123-128)
#  384 B  one accepted back-transformed hit          (three Matrix4x4)
#   96 B  per ray                                    (ray in + hit record out)
# The work counts are those of the reference's algorithm, produced by its CPU restatement
# (oracle/rt_oracle.c) in an untimed pass; the device's own ray and triangle-test counters must
# agree with them (reported as "counters_agree").
HBM_PEAK_GBS = 8000.0


class OracleCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "primary_rays", "shadow_rays", "secondary_rays", "node_pops",
                                          "tri_tests", "world_box_tests", "ray_transforms", "transformed_hits",
                                          "shades")]


def algorithmic_bytes(c) -> int:
    return (72 * c["node_pops"] + 88 * c["tri_tests"] + 48 * c["world_box_tests"] + 128 * c["ray_transforms"]
            + 384 * c["transformed_hits"] + 96 * c["rays"])


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


def reference_work_counts(desc, cam, nthreads):
    """Work counters of the reference's algorithm for one frame, from its CPU restatement
    (untimed; None when the checker library is not built)."""
    so = os.path.join(ROOT, "oracle", "librt_oracle.so")
    if not os.path.exists(so):
        return None, None
    from raytracer_amd import _abi
    lib = C.CDLL(so)
    lib.rt_oracle_render.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.Camera), C.c_void_p, C.c_void_p,
                                     C.POINTER(OracleCounters), C.c_int]
    lib.rt_oracle_render.restype = C.c_double
    rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
    ctr = OracleCounters()
    ms = lib.rt_oracle_render(desc, C.byref(cam), rgb.ctypes.data, None, C.byref(ctr), nthreads)
    if ms < 0:
        return None, None
    return {k: int(getattr(ctr, k)) for k, _ in OracleCounters._fields_}, ms


def scene_cache_dir():
    d = os.environ.get("RTX_BENCH_SCENES", os.path.join(tempfile.gettempdir(), f"rtx_bench_scenes_{os.getuid()}"))
    os.makedirs(d, exist_ok=True)
    return d


def make_workload(name: str, d: str, frame: int = 0, n_frames: int = 360):
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
    if name == "metal_glass_plates":
        p = os.path.join(d, "metal_glass_plates.json")
        if not os.path.exists(p):
            scenes.scene_plates(d, res=(800, 800), name="metal_glass_plates")
        return p, "metal_glass_plates stand-in: plane, spheres, dielectric boxes of free triangles, 800x800"
    if name == "camera_around_david":
        p = os.path.join(d, f"davids_camera_{frame:03d}.json")
        if not os.path.exists(p):
            scenes.scene_david_frame(d, frame, n_frames=n_frames, faces=100_000, res=(800, 800), name="davids_camera")
        return p, "camera_around_david stand-in: 100k-face mesh + mirror ground, 800x800, 360-frame orbit"
    raise SystemExit(f"unknown workload {name}")


def orbit_camera(cam, angle_deg: float):
    """Rotate the camera about the world y axis through the origin (whole-frame sharding
    gives every rank its own frame of an orbit around the same scene)."""
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


def cpu_reference_baseline(scene_json: str, rays: int, sample_f: int = 1):
    """Time the unmodified reference renderer (oracle/_ref/raytracer_ref, built from the
    reference's own sources) on this host: `./raytracer <scene> -b`, multithreaded default
    = min(4, nproc) threads (raytracer.cpp:99-100).  Reads renderTimeMs from its results
    JSON (raytracer.cpp:269-282).  sample_f > 1: bounded sample, see cpu_sample_factor."""
    ref = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
    if not os.access(ref, os.X_OK):
        return None
    work = tempfile.mkdtemp(prefix="rtx_ref_")
    try:
        os.makedirs(os.path.join(work, "run"))
        os.makedirs(os.path.join(work, "my_outputs_hw2"))
        if sample_f > 1:
            scene_json = reduced_scene_file(scene_json, sample_f)
        t0 = time.time()
        r = subprocess.run([ref, scene_json, "-b"], cwd=os.path.join(work, "run"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=900)
        wall = time.time() - t0
        if r.returncode != 0:
            return None
        res = [f for f in os.listdir(os.path.join(work, "my_outputs_hw2")) if f.endswith("_results.json")]
        if not res:
            return None
        with open(os.path.join(work, "my_outputs_hw2", res[0])) as f:
            j = json.load(f)
        render_ms = float(j["renderTimeMs"]) * sample_f * sample_f
        sample = "whole frame once" if sample_f == 1 else \
            f"the same view at 1/{sample_f} of the resolution in both directions, time scaled by {sample_f * sample_f}"
        return {"value": rays / render_ms / 1e3, "unit": "Mrays/s", "cores": min(4, os.cpu_count() or 4),
                "kind": "reference", "ms_per_frame": render_ms, "preprocess_ms": float(j["preprocessingTimeMs"]),
                "host_cpus": os.cpu_count(),
                "sample": f"{sample}, ./raytracer <scene> -b (reference caps itself at 4 threads); process wall {wall:.1f} s"}
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dragon_metal")
    ap.add_argument("--mode", choices=["frames", "tiles"], default="frames")
    ap.add_argument("--tile", default="64x8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--consistent", action="store_true",
                    help="time the library's consistent mode (NOT the reference's results for scaled instances; "
                         "never the headline number).  Rays and algorithmic bytes stay those of the same frame in "
                         "the default, reference-exact mode")
    args = ap.parse_args()

    import torch
    import raytracer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU path behind the C-ABI")
    # Rehearsal on a one-GPU box (RTX_BENCH_REHEARSAL=1): every rank uses device 0 and the
    # collectives run over gloo on host tensors; everything else is the multi-GPU code path.
    rehearsal = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearsal else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    d = scene_cache_dir()
    if rank == 0:
        path, descr = make_workload(args.workload, d)
    barrier()
    path, descr = make_workload(args.workload, d)

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"workload {args.workload}: loading scene")
    t0 = time.time()
    hs = rt.HostScene(path)
    t_load = time.time() - t0
    t0 = time.time()
    ds = rt.DeviceScene(hs.desc, local_rank)
    t_upload = time.time() - t0
    cam0, _ = hs.camera(0)
    tw, th = (int(x) for x in args.tile.split("x"))

    if args.mode == "frames":
        cam = orbit_camera(cam0, 360.0 * rank / max(world, 1) if world > 1 else 0.0)
    else:
        cam = cam0

    # ---- untimed counting pass: rays and traversal work of this rank's step ----------------
    note("device counting pass")
    ds.set_counters(True)
    if args.mode == "frames":
        _, _, cst = ds.render(cam)
    else:
        _, cst = ds.render_tiles(cam, tw, th, rank, world)
    ds.set_counters(False)
    rays_rank = int(cst["rays"])
    host_threads = max(1, min(16, os.cpu_count() or 1) // max(world, 1)) if world > 1 else min(16, os.cpu_count() or 1)
    ref_counts, oracle_ms = (None, None)
    sample_f = cpu_sample_factor(hs.desc, cam)
    note(f"reference work counts on the host (sample factor {sample_f})")
    if args.mode == "frames":
        ref_counts, oracle_ms = reference_work_counts(hs.desc, cam if sample_f == 1 else reduced_camera(cam, sample_f),
                                                      host_threads)
    if ref_counts is not None and sample_f > 1:
        # bounded sample: the same view at 1/f of the resolution in both directions, counts scaled
        # to the full frame by the device's own ray count
        scale = rays_rank / max(ref_counts["rays"], 1)
        ref_counts = {k: int(round(v * scale)) for k, v in ref_counts.items()}
        counters_agree = None
        alg_bytes_rank = algorithmic_bytes(ref_counts)
        alg_source = (f"reference algorithm's work counts (CPU restatement on the same view at 1/{sample_f} resolution, "
                      "scaled to the frame's ray count) x reference record sizes")
    elif ref_counts is not None:
        # (world-box tests are not compared: the device's TLAS rejects most instances before
        # their boxes are tested, the reference tests every instance's box, utils.cpp:756-766)
        counters_agree = ref_counts["rays"] == cst["rays"] and ref_counts["tri_tests"] == cst["tri_tests"]
        alg_bytes_rank = algorithmic_bytes(ref_counts)
        alg_source = "reference algorithm's work counts (CPU restatement, untimed pass) x reference record sizes"
    else:
        # tiles mode / checker not built: the device's counters, a collapsed-node visit priced as the
        # three binary nodes it stands for at most
        counters_agree = None
        alg_bytes_rank = algorithmic_bytes({"node_pops": 3 * cst["node_visits"], "tri_tests": cst["tri_tests"],
                                            "world_box_tests": cst["world_box_tests"],
                                            "ray_transforms": cst["object_space_traversals"],
                                            "transformed_hits": 0, "rays": cst["rays"]})
        alg_source = "device counters (estimate)"

    # the caller-owned image buffer of the C-ABI: page-locked and reused from frame to frame
    frame = torch.empty((cam.height, cam.width, 3), dtype=torch.uint8).pin_memory().numpy()
    gather_buf = None
    my_bytes = ds.tile_bytes(cam, tw, th, rank, world) if args.mode == "tiles" else 0
    if args.mode == "tiles":
        sizes = [ds.tile_bytes(cam, tw, th, r, world) for r in range(world)]
        pad = max(sizes)
        my_dev = torch.empty(pad, dtype=torch.uint8, device="cuda")
        if rank == 0 and world > 1:
            gather_buf = [torch.empty(pad, dtype=torch.uint8, device=red_dev) for _ in range(world)]

    def step():
        if args.mode == "frames":
            _, _, st = ds.render(cam, out=frame)
            return st
        _, st = ds.render_tiles(cam, tw, th, rank, world, out_ptr=my_dev.data_ptr(), out_is_device=True)
        if world > 1:
            dist.gather(my_dev.cpu() if rehearsal else my_dev, gather_buf, dst=0)
        if rank == 0:
            parts = gather_buf if world > 1 else [my_dev]
            for r, t in enumerate(parts):
                rt.unpack_tiles(t[:sizes[r]].cpu().numpy(), cam.width, cam.height, tw, th, r, world, frame)
        return st

    if args.consistent:
        ds.set_consistent(True)
    note("timed steps")
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    trace_ms = 0.0
    trace_launches = 0
    for _ in range(args.steps):
        st = step()
        kernel_ms += st["kernel_ms"]
        trace_ms += st["trace_ms"]
        trace_launches += st["trace_launches"]
    barrier()
    elapsed = time.perf_counter() - t0

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(rays_rank), float(alg_bytes_rank)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_step = float(tot[0].item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_step * args.steps / elapsed / 1e6
        # Dominant kernel: k_trace, one launch per ray generation.  Its launches are bracketed by HIP
        # events on the library's own stream (rtx_stats.trace_ms / trace_launches).  A launch
        # processes, on average, 1/launches_per_frame of the frame's algorithmic bytes.
        launches_per_frame = trace_launches / args.steps
        avg_launch_ms = trace_ms / max(trace_launches, 1)
        bytes_per_launch = alg_bytes_rank / max(launches_per_frame, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        k_ms_frame = kernel_ms / args.steps                    # every kernel of rank 0's frame
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                with open(tfile) as f:
                    tj = json.load(f)
                if tj.get("workload") == args.workload:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (ms/frame in ms_per_step)", "value": mrays, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if args.mode == "frames" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "description": descr, "mode": args.mode,
                       "resolution": [cam.width, cam.height], "rays_per_step": rays_step,
                       "rays_per_pixel": rays_rank / (cam.width * cam.height) if args.mode == "frames" else None,
                       "frames_per_step": world if args.mode == "frames" else 1,
                       "host_load_s": t_load, "scene_upload_s": t_upload,
                       "image_buffer": "host, page-locked, reused (download inside the timed step)",
                       "traversal": ("consistent mode: not the reference's results where instances are scaled; rays and "
                                     "algorithmic bytes are those of the same frame in the default mode")
                                    if args.consistent else "reference-exact (default)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_trace", "launches_per_frame": launches_per_frame,
                         "avg_launch_ms": avg_launch_ms, "trace_ms_per_frame": trace_ms / args.steps,
                         "all_kernels_ms_per_frame": k_ms_frame,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_frame": alg_bytes_rank,
                         "algorithmic_bytes_per_ray": alg_bytes_rank / max(rays_rank, 1),
                         "algorithmic_source": alg_source, "counters_agree": counters_agree,
                         "note": ("algorithmic bytes are those of the reference's algorithm (its work counts x its record "
                                  "sizes); the device may undercut them with result-identical shortcuts (TLAS instead of the "
                                  "linear instance loop, early retirement of occluded shadow rays, caches), so on workloads "
                                  "where the reference re-reads a small working set per ray frac can exceed 1"),
                         "reference_work_counts": ref_counts,
                         "device_record_visits": int(cst["node_visits"]), "tri_tests": int(cst["tri_tests"]),
                         "device_world_box_tests": int(cst["world_box_tests"]),
                         "device_object_space_traversals": int(cst["object_space_traversals"]),
                         "device_tlas_node_visits": int(cst["tlas_node_visits"]),
                         "lanes_at_nodes_per_wave_step": cst["wave_inner_lanes"] / max(cst["wave_inner_steps"], 1),
                         "lanes_at_leaves_per_wave_step": cst["wave_leaf_lanes"] / max(cst["wave_leaf_steps"], 1),
                         "wave_steps_with_node_lanes": int(cst["wave_inner_steps"]),
                         "wave_steps_with_leaf_lanes": int(cst["wave_leaf_steps"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            note("reference CPU renderer")
            cb = cpu_reference_baseline(path, rays_rank, sample_f)
            if cb is not None:
                out["cpu_baseline"] = cb
                out["config"]["speedup_vs_cpu_reference_ms_per_frame"] = cb["ms_per_frame"] / ms_per_step
        print(json.dumps(out), flush=True)
    ds.close()
    hs.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
