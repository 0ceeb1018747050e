"""The CPU checker of the device library's consistent mode (rt_oracle_render_consistent).
That mode has no counterpart in the reference, so there is nothing of the reference's to pin it
to; what can be checked is that it coincides with the reference-mode restatement wherever the two
must agree, and that it differs where the reference's transformed-mesh code loses or invents
hits (utils.cpp:476-531, 559-570)."""
import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt

INSTANCE_COLOUR = np.array([25, 50, 200], dtype=np.uint8)      # ambient-only material of the pyramid scenes


def both_modes(path):
    hs = rt.HostScene(path)
    cam, _ = hs.camera(0)
    a = ck.oracle_render(hs.desc, cam)
    b = ck.oracle_render(hs.desc, cam, consistent=True)
    hs.close()
    return a, b


def instance_pixels(rgb):
    return int(np.sum(np.all(rgb.reshape(-1, 3) == INSTANCE_COLOUR, axis=1)))


@pytest.mark.parametrize("name", ["unit", "magnified", "buried_unit", "bumpy_small", "david_small"])
def test_modes_agree_without_shrunk_or_buried_instances(unit_scenes, name):
    (rgb, hits, ctr, _), (c_rgb, c_hits, c_ctr, _) = both_modes(unit_scenes[name])
    assert np.array_equal(rgb, c_rgb) and np.array_equal(hits, c_hits)
    assert ctr["rays"] == c_ctr["rays"]


def test_shrunk_instance_is_found():
    """Scale 0.25: object-space distances are four times the world ones, so the reference never
    finds the pyramid behind the bound the ground left (0 pixels); it is there."""
    import tempfile
    from raytracer_amd import scenes
    with tempfile.TemporaryDirectory() as d:
        (rgb, _, _, _), (c_rgb, c_hits, _, _) = both_modes(scenes.scene_q1(d, "shrunk"))
    assert instance_pixels(rgb) == 0
    assert instance_pixels(c_rgb) > 0
    assert int(np.sum(c_hits["container"] == -2)) == instance_pixels(c_rgb)


def test_buried_instance_leaves_no_trace():
    """Scale 2, almost all of the pyramid under the ground: the reference drops the hits below the
    ground but keeps their records (144 pixels in the instance's colour); nothing of it shows."""
    import tempfile
    from raytracer_amd import scenes
    with tempfile.TemporaryDirectory() as d:
        (rgb, _, _, _), (c_rgb, c_hits, _, _) = both_modes(scenes.scene_q2(d, "buried"))
        (u_rgb, _, _, _), _ = both_modes(scenes.scene_q2(d, "buried_unit"))
    assert instance_pixels(rgb) == 144
    assert instance_pixels(c_rgb) <= 4 * max(1, instance_pixels(u_rgb))
    assert int(np.sum(c_hits["container"] == -2)) == instance_pixels(c_rgb)
