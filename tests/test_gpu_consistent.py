"""The consistent mode of the HIP core (rtx_scene_set_consistent) against its CPU checker.
The mode departs from the reference on purpose (scaled instances, any-hit shadow rays), so the
comparison is with rt_oracle_render_consistent, to the same bar as the parity tests."""
import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

UNIT = list(scenes.UNIT_SCENES.keys())


@pytest.mark.parametrize("name", UNIT)
def test_consistent_mode_matches_its_checker(unit_scenes, name):
    hs = rt.HostScene(unit_scenes[name])
    ds = rt.DeviceScene(hs.desc, 0)
    ds.set_consistent(True)
    for k in range(hs.n_cameras):
        cam, _ = hs.camera(k)
        rgb, hits, _ = ds.render(cam, want_hits=True)
        o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, cam, consistent=True)
        bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
        assert len(bad) == 0, f"{name}: {len(bad)} primary hits differ, first {bad[:5]}: {hits[bad[:3]]} vs {o_hits[bad[:3]]}"
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= 2 and same >= 0.999, (name, same, mx)
    ds.close(); hs.close()


def test_mode_switch_is_stateless_and_counters_are_refused(unit_scenes):
    hs = rt.HostScene(unit_scenes["marching_small"])
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    first, first_hits, _ = ds.render(cam, want_hits=True)
    ds.set_consistent(True)
    other, _, _ = ds.render(cam, want_hits=True)
    ds.set_counters(True)
    with pytest.raises(rt.RtxError):
        ds.render(cam)
    ds.set_counters(False)
    ds.set_consistent(False)
    again, again_hits, _ = ds.render(cam, want_hits=True)
    assert np.array_equal(first, again) and np.array_equal(first_hits, again_hits)
    assert not np.array_equal(first, other)            # this scene has scaled instances
    ds.close(); hs.close()


def test_same_picture_as_default_mode_without_transformed_meshes(tmp_path):
    """No transformed mesh or instance: both modes must give the same picture (an any-hit shadow
    query and the reference's closest-hit query answer alike), at a size with deep trees."""
    path = scenes.scene_david_frame(str(tmp_path), 3, n_frames=12, faces=20000, res=(320, 240))
    hs = rt.HostScene(path)
    ds = rt.DeviceScene(hs.desc, 0)
    cam, _ = hs.camera(0)
    a, ah, _ = ds.render(cam, want_hits=True)
    ds.set_consistent(True)
    b, bh, _ = ds.render(cam, want_hits=True)
    assert np.array_equal(ah, bh)
    same, mx = ck.image_diff(a, b)
    assert mx <= 1 and same >= 0.9999, (same, mx)
    ds.close(); hs.close()
