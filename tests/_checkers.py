"""Test-side access to the checkers: the CPU restatement (oracle/librt_oracle.so) and,
where it was built, the compiled reference (oracle/_ref/).  Only tests, smoke() and the
cpu_baseline leg of bench.py import this module; the product never does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_DUMP = os.path.join(REF_DIR, "ref_dump")
REF_BIN = os.path.join(REF_DIR, "raytracer_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class OracleCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("rays", "primary_rays", "shadow_rays", "secondary_rays", "node_pops", "tri_tests",
                 "world_box_tests", "ray_transforms", "transformed_hits", "shades",
                 "tri_past_det", "tri_past_beta", "tri_past_gamma")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        lib.rt_oracle_render.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.Camera), C.c_void_p,
                                         C.c_void_p, C.POINTER(OracleCounters), C.c_int]
        lib.rt_oracle_render.restype = C.c_double
        lib.rt_oracle_render_consistent.argtypes = lib.rt_oracle_render.argtypes
        lib.rt_oracle_render_consistent.restype = C.c_double
        _oracle = lib
    return _oracle


def oracle_render(desc, cam, want_hits=True, nthreads=8, consistent=False):
    """-> (rgb uint8[h][w][3], hits structured array | None, counters dict, elapsed ms)
    consistent: the restatement of the device library's consistent mode instead of the reference's."""
    rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
    hits = np.empty(cam.height * cam.width, dtype=_abi.HIT_DTYPE) if want_hits else None
    ctr = OracleCounters()
    fn = oracle_lib().rt_oracle_render_consistent if consistent else oracle_lib().rt_oracle_render
    ms = fn(desc, C.byref(cam), rgb.ctypes.data, hits.ctypes.data if want_hits else None, C.byref(ctr), nthreads)
    if ms < 0:
        raise RuntimeError("rt_oracle_render rejected its arguments")
    return rgb, hits, ctr.as_dict(), ms


def have_reference():
    return os.access(REF_DUMP, os.X_OK)


def reference_dump(json_path, out_prefix, cull=False, bvh=True, render=True):
    """Run the reference-linked dumper; returns list of (rgb, hits) per camera."""
    cmd = [REF_DUMP, json_path, out_prefix, "--cull", "1" if cull else "0", "--bvh", "1" if bvh else "0"]
    if not render:
        cmd.append("--no-render")
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_dump failed: " + r.stderr[-2000:])
    return r.stderr


def read_rgb(path, w, h):
    return np.fromfile(path, dtype=np.uint8).reshape(h, w, 3)


def read_hits(path):
    return np.fromfile(path, dtype=_abi.HIT_DTYPE)


def read_sections(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        name = data[pos:pos + 24].split(b"\0", 1)[0].decode()
        n = int.from_bytes(data[pos + 24:pos + 32], "little")
        out[name] = data[pos + 32:pos + 32 + n]
        pos += 32 + n
    return out


def compare_hits(a, b, allow_src_unknown=True):
    """Indices of pixels whose primary-hit identity differs (kind, container, face, src,
    distance bits).  A src of -1 in `b` (the reference-side attribution can fail for
    records the reference itself left inconsistent) matches anything."""
    bad = (a["kind"] != b["kind"]) | (a["container"] != b["container"]) | (a["face"] != b["face"])
    bad |= a["distance"].view(np.uint64) != b["distance"].view(np.uint64)
    src_bad = a["src"] != b["src"]
    if allow_src_unknown:
        src_bad &= ~((b["src"] == -1) & (b["kind"] == _abi.KIND_MESH))
    bad |= src_bad
    return np.nonzero(bad)[0]


def image_diff(a, b):
    """-> (fraction of pixels identical, max per-channel abs difference)"""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    same = (d.reshape(-1, 3).max(axis=1) == 0).mean()
    return float(same), int(d.max()) if d.size else 0
