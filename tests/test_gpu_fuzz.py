"""Seeded random scenes (scenes.scene_random) on the GPU: the HIP core against the CPU
restatement in its three modes -- counters on (every test of the reference carried out), the
default mode, and the consistent mode against its own checker -- and, where a scene qualifies, the
default and consistent modes again over the eight-wide float-box records (rtx_scene_set_wide_blas)."""
import os

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu

# RTX_FUZZ_SEEDS=N widens the sweep (a soak run; the default keeps the suite short)
SEEDS = sorted(set(range(int(os.environ.get("RTX_FUZZ_SEEDS", "64")))) | {110})     # 110: a rejected transformed free triangle in front of a light


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_matches_oracle(tmp_path, seed):
    path = scenes.scene_random(str(tmp_path), seed)
    cull = bool(seed & 1)
    hs = rt.HostScene(path, True, cull)
    ds = rt.DeviceScene(hs.desc, 0, wide_blas=True)
    cam, _ = hs.camera(0)
    o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, cam)
    for counters in (True, False):
        ds.set_counters(counters)
        rgb, hits, st = ds.render(cam, want_hits=True)
        bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
        assert len(bad) == 0, (seed, counters, bad[:5], hits[bad[:3]], o_hits[bad[:3]])
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= 2 and same >= 0.999, (seed, counters, same, mx)
        if counters:
            assert st["rays"] == o_ctr["rays"] and st["tri_tests"] == o_ctr["tri_tests"], (seed, st, o_ctr)
    # (the default-mode frame above settled its free shadow rays in the shading kernel; once with all of them traced)
    ds.set_debug(_abi.DEBUG_NO_EARLY_SHADOW)
    q_rgb, q_hits, q_st = ds.render(cam, want_hits=True)
    ds.set_debug(0)
    assert np.array_equal(q_hits, hits) and np.array_equal(q_rgb, rgb) and q_st["rays"] == st["rays"], (seed, "every shadow ray traced")
    ds.set_wide_blas(True)
    w_rgb, w_hits, w_st = ds.render(cam, want_hits=True)
    assert np.array_equal(w_hits, hits) and np.array_equal(w_rgb, rgb), (seed, "eight-wide records")
    ds.set_consistent(True)
    c_rgb, c_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=True)
    for wide in (False, True):
        ds.set_wide_blas(wide)
        rgb, hits, _ = ds.render(cam, want_hits=True)
        assert len(ck.compare_hits(hits, c_hits, allow_src_unknown=False)) == 0, (seed, wide)
        same, mx = ck.image_diff(rgb, c_rgb)
        assert mx <= 2 and same >= 0.999, (seed, "consistent", wide, same, mx)
    ds.close(); hs.close()


@pytest.mark.parametrize("seed", [1, 2, 6, 9, 13, 21, 34, 55])
def test_detailed_random_scene_matches_oracle(tmp_path, seed):
    """The same with meshes of 10-40 thousand faces at 288x208: traversals long enough for the
    wave-wide finishing of drained waves and for the tiny-generation kernel."""
    path = scenes.scene_random(str(tmp_path), seed, res=(288, 208), detail=12)
    hs = rt.HostScene(path, True, bool(seed & 1))
    ds = rt.DeviceScene(hs.desc, 0, wide_blas=True)
    cam, _ = hs.camera(0)
    used_wide = 0
    for consistent in (False, True):
        ds.set_consistent(consistent)
        o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=consistent)
        for wide in (False, True):
            ds.set_wide_blas(wide)
            rgb, hits, st = ds.render(cam, want_hits=True)
            used_wide += st["wide_blas"]
            assert st["wide_blas_redone"] == 0
            assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0, (seed, consistent, wide)
            same, mx = ck.image_diff(rgb, o_rgb)
            assert mx <= 2 and same >= 0.999, (seed, consistent, wide, same, mx)
    assert used_wide in (0, 2)          # (scenes with more than four instances keep the four-wide records)
    ds.close(); hs.close()
