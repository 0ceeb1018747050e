"""Exact mode's nearest-first search of the instances (scenes with 32 instances or more) against
the CPU restatement and against the plain replay in index order that it stands in for: crowded
random scenes (instances that copy one another, untransformed ones, scales below and above one,
mirrored ones, instances of instances) and the grass stand-ins.  Hit records and picture bytes must
be equal: the search is an evaluation order, not an approximation."""
import os

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu


def _both_ways(ds, cam):
    ds.set_debug(0)
    a = ds.render(cam, want_hits=True)
    ds.set_debug(_abi.DEBUG_NO_SEARCH)             # the index-order replay the search stands in for
    try:
        b = ds.render(cam, want_hits=True)
    finally:
        ds.set_debug(0)
    return a, b


# RTX_SEARCH_SEEDS=N widens the sweep (a soak run; the default keeps the suite short)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RTX_SEARCH_SEEDS", "24")))))
def test_crowded_random_scene(tmp_path, seed):
    path = scenes.scene_random(str(tmp_path), seed, res=(96, 72), crowd=40 + 12 * (seed % 9))
    hs = rt.HostScene(path, True, bool(seed & 1))
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    (rgb, hits, _), (r_rgb, r_hits, _) = _both_ways(ds, cam)
    assert np.array_equal(hits, r_hits) and np.array_equal(rgb, r_rgb), seed
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam)
    bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
    assert len(bad) == 0, (seed, bad[:5], hits[bad[:3]], o_hits[bad[:3]])
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999, (seed, same, mx)
    # the consistent mode's nearest-first walk (taken from 32 instances on) against its own checker
    ds.set_consistent(True)
    c_rgb, c_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=True)
    rgb, hits, st = ds.render(cam, want_hits=True)
    assert st["instance_path"] == 2
    assert len(ck.compare_hits(hits, c_hits, allow_src_unknown=False)) == 0, (seed, "consistent")
    same, mx = ck.image_diff(rgb, c_rgb)
    assert mx <= 2 and same >= 0.999, (seed, "consistent", same, mx)
    ds.close(); hs.close()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("RTX_SEARCH_DETAILED", "2,5,11").split(",")])
def test_crowded_detailed_random_scene(tmp_path, seed):
    """Meshes of several thousand faces under 100-160 instances at 224x160: deep BLAS traversals
    inside the search, long lists of found instances."""
    path = scenes.scene_random(str(tmp_path), seed, res=(224, 160), detail=6, crowd=100 + 10 * (seed % 7))
    hs = rt.HostScene(path, True, bool(seed & 1))
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    (rgb, hits, _), (r_rgb, r_hits, _) = _both_ways(ds, cam)
    assert np.array_equal(hits, r_hits) and np.array_equal(rgb, r_rgb), seed
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam)
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0, seed
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999, (seed, same, mx)
    ds.close(); hs.close()


@pytest.mark.parametrize("name", ["grass_small", "grass_many"])
def test_grass_search_equals_replay(unit_scenes, name):
    hs = rt.HostScene(unit_scenes[name])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    (rgb, hits, _), (r_rgb, r_hits, _) = _both_ways(ds, cam)
    assert np.array_equal(hits, r_hits) and np.array_equal(rgb, r_rgb)
    ds.close(); hs.close()


def test_dense_grass_at_size(tmp_path):
    """4000 instances seen at 640x360: long walks, many overwritten records (scales up to 2.2 put
    blades behind the nearest one in front of it by the reference's comparison)."""
    path = scenes.scene_grass(str(tmp_path), n_instances=4000, res=(640, 360), name="grass_mid", seed=77)
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    (rgb, hits, st), (r_rgb, r_hits, r_st) = _both_ways(ds, cam)
    assert np.array_equal(hits, r_hits) and np.array_equal(rgb, r_rgb)
    # the search is the same picture from fewer records (rtx_stats.record_fetches is counted in every mode)
    assert 0 < st["record_fetches"] < 0.8 * r_st["record_fetches"], (st["record_fetches"], r_st["record_fetches"])
    assert st["rays"] == r_st["rays"]
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam)
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= 2 and same >= 0.999, (same, mx)
    ds.close(); hs.close()


def test_a_scene_that_mostly_falls_back_is_left_to_the_replay(tmp_path):
    """rtx_stats.instance_path: the search on the first frame; the replay from the second on when more
    than one ray in ten fell back (here: scales on both sides of one); grass keeps the search."""
    path = scenes.scene_random(str(tmp_path), 3, res=(96, 72), crowd=76)
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    first, _, st1 = ds.render(cam)
    second, _, st2 = ds.render(cam)
    assert st1["instance_path"] == 1 and st2["instance_path"] == 0
    assert np.array_equal(first, second)
    ds.set_consistent(True)
    assert ds.render(cam)[2]["instance_path"] == 2
    ds.close(); hs.close()
    grass = scenes.scene_grass(str(tmp_path), n_instances=400, res=(96, 54), name="grass_path")
    hs = rt.HostScene(grass)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    assert [ds.render(cam)[2]["instance_path"] for _ in range(3)] == [1, 1, 1]
    ds.close(); hs.close()
