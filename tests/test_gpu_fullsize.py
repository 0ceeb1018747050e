"""GPU parity at the sizes the benchmark runs: the full-size scenes (1.8 M-face mesh with a
deep BVH; 10 000 instances) rendered by the HIP core through its C-ABI.

The CPU oracle cannot render a full frame of these in test time, so two kinds of check
are used:

* the full-size *frame* itself against the CPU restatement on all host threads and against the
  picture the compiled reference writes for the same scene file, for the workloads whose
  reference frame takes seconds (`test_full_frame_matches_oracle_and_reference`);
* the full-size *scene* on a reduced view (same camera, image reduced by an integer
  factor): every primary hit must equal the oracle's bit for bit, the image must meet the
  parity bar, and the closest-hit query counts must agree;
* size-independent properties of the full-size *frame*: interleaved tile subsets (the
  multi-GPU decomposition) reassemble to the whole frame, the frame is repeatable, every
  primary-hit record is well formed, and the reduced view is consistent with it.
"""
import numpy as np
import pytest

import _checkers as ck
import bench
import raytracer_amd as rt

pytestmark = pytest.mark.gpu

MIN_SAME = 0.999
MAX_DELTA = 2


@pytest.fixture(scope="module")
def workloads():
    cache = {}

    def get(name):
        if name not in cache:
            path, _ = bench.make_workload(name, bench.scene_cache_dir())
            hs = rt.HostScene(path)
            ds = rt.DeviceScene(hs.desc, 0)
            cache[name] = (hs, ds)
        return cache[name]

    yield get
    for hs, ds in cache.values():
        ds.close()
        hs.close()


@pytest.mark.parametrize("name,factor", [("dragon_metal", 3), ("marching_dragons", 3), ("grass_desert", 8),
                                         ("metal_glass_plates", 2), ("camera_around_david", 2)])
def test_full_size_scene_matches_oracle_on_reduced_view(workloads, name, factor):
    hs, ds = workloads(name)
    cam, _ = hs.camera(0)
    small = bench.reduced_camera(cam, factor)
    ds.set_counters(True)
    rgb, hits, st = ds.render(small, want_hits=True)
    ds.set_counters(False)
    o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, small)
    bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
    assert len(bad) == 0, f"{name}: {len(bad)} primary hits differ, e.g. {hits[bad[:3]]} vs {o_hits[bad[:3]]}"
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= MAX_DELTA and same >= MIN_SAME, f"{name}: {same:.5f} identical, max delta {mx}"
    for k in ("rays", "primary_rays", "shadow_rays", "secondary_rays", "tri_tests"):
        assert st[k] == o_ctr[k], (k, st[k], o_ctr[k])
    # the default mode (counters off: occluded shadow rays retire early) gives the same results
    rgb2, hits2, _ = ds.render(small, want_hits=True)
    assert len(ck.compare_hits(hits2, o_hits, allow_src_unknown=False)) == 0
    same, mx = ck.image_diff(rgb2, o_rgb)
    assert mx <= MAX_DELTA and same >= MIN_SAME, f"{name} (default mode): {same:.5f} identical, max delta {mx}"


@pytest.mark.parametrize("name,factor", [("dragon_metal", 1), ("marching_dragons", 1), ("metal_glass_plates", 1),
                                         ("camera_around_david", 1), ("grass_desert", 4)])
def test_full_frame_matches_oracle_and_reference(workloads, name, factor):
    """The frame the benchmark times (grass_desert: the same view at 1/4 of the resolution in
    both directions -- the reference's linear instance loop needs minutes for the 4K frame):
    every primary-hit record equals the oracle's, the picture meets the parity bar against the
    oracle and against the PPM the unmodified reference renderer writes (raytracer.cpp:45-51,78-80)."""
    import os
    threads = min(16, os.cpu_count() or 1)
    hs, ds = workloads(name)
    cam, _ = hs.camera(0)
    view = bench.reduced_camera(cam, factor) if factor > 1 else cam
    rgb, hits, st = ds.render(view, want_hits=True)
    o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, view, nthreads=threads)
    bad = ck.compare_hits(hits, o_hits, allow_src_unknown=False)
    assert len(bad) == 0, f"{name}: {len(bad)} primary hits differ"
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= MAX_DELTA and same >= MIN_SAME, f"{name}: {same:.5f} identical, max delta {mx}"
    assert st["rays"] == o_ctr["rays"]            # (ray counts need no device counters)
    if os.access(ck.REF_BIN, os.X_OK):
        _, ref_img = bench.cpu_reference_baseline(hs.path, 1, sample_f=factor, runs=1)
        assert ref_img is not None and ref_img.shape == rgb.shape
        same, mx = ck.image_diff(rgb, ref_img)
        assert mx <= MAX_DELTA and same >= MIN_SAME, f"{name} vs reference PPM: {same:.5f} identical, max delta {mx}"
        assert np.array_equal(o_rgb, ref_img), "the oracle's frame is the reference's, byte for byte"


@pytest.mark.parametrize("name,factor", [("grass_desert", 5), ("marching_dragons", 2)])
def test_consistent_mode_at_size(workloads, name, factor):
    """Consistent mode on the instance workloads (its nearest-first walk of the instances) against
    its own CPU checker: every primary-hit record equal, the picture to the parity bar."""
    import os
    hs, ds = workloads(name)
    cam, _ = hs.camera(0)
    view = bench.reduced_camera(cam, factor)
    ds.set_consistent(True)
    try:
        rgb, hits, st = ds.render(view, want_hits=True)
    finally:
        ds.set_consistent(False)
    o_rgb, o_hits, _, _ = ck.oracle_render(hs.desc, view, consistent=True, nthreads=min(16, os.cpu_count() or 1))
    assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0, name
    same, mx = ck.image_diff(rgb, o_rgb)
    assert mx <= MAX_DELTA and same >= MIN_SAME, f"{name} (consistent): {same:.5f} identical, max delta {mx}"
    # (the nearest-first walk is taken from 32 instances on; marching_dragons, with eight, replays them)
    assert st["instance_path"] == (2 if name == "grass_desert" else 0)


@pytest.mark.parametrize("name,tile", [("dragon_metal", (64, 64)), ("grass_desert", (128, 72))])
def test_full_frame_properties(workloads, name, tile):
    hs, ds = workloads(name)
    cam, _ = hs.camera(0)
    full, hits, st = ds.render(cam, want_hits=True)
    assert st["queue_overflows"] == 0 or st["generations"] > 0
    # well-formed primary-hit records
    d = hs.desc.contents
    kinds = hits["kind"]
    assert kinds.min() >= 0 and kinds.max() <= 4
    mesh_hits = hits[kinds == rt._abi.KIND_MESH]
    assert mesh_hits["src"].min() >= 0 and mesh_hits["src"].max() < d.n_meshes + d.n_instances
    assert mesh_hits["face"].min() >= 0
    assert np.all(np.isfinite(mesh_hits["distance"])) and mesh_hits["distance"].min() > 0.0
    assert np.all(hits["distance"][kinds == 0] == 0.0)
    # repeatable: same hit records, same bytes (pixel sums are accumulated in fixed point, whatever the order)
    again, hits2, _ = ds.render(cam, want_hits=True)
    assert np.array_equal(hits, hits2)
    same, mx = ck.image_diff(full, again)
    assert np.array_equal(full, again)
    # the multi-GPU decomposition: interleaved tile subsets reassemble to the frame
    tw, th = tile
    for world in (2, 8):
        frame = np.zeros_like(full)
        for rank in range(world):
            packed, _ = ds.render_tiles(cam, tw, th, rank, world)
            rt.unpack_tiles(packed, cam.width, cam.height, tw, th, rank, world, frame)
        same, mx = ck.image_diff(frame, full)
        assert np.array_equal(frame, full), f"{name}: tiles x{world}: {same:.6f} identical, max delta {mx}"


def test_grass_desert_search_equals_replay_at_4k(workloads):
    """The default path of the 4K frame the benchmark times -- 10 000 instances searched nearest
    first with verification -- against the index-order replay of the reference's loop
    (utils.cpp:756-790) it stands in for: every hit record and every byte of the picture."""
    from raytracer_amd import _abi
    hs, ds = workloads("grass_desert")
    cam, _ = hs.camera(0)
    assert (cam.width, cam.height) == (3840, 2160) and hs.desc.contents.n_instances == 10_000
    ds.set_debug(0)
    rgb, hits, st = ds.render(cam, want_hits=True)
    ds.set_debug(_abi.DEBUG_NO_SEARCH)
    try:
        r_rgb, r_hits, r_st = ds.render(cam, want_hits=True)
    finally:
        ds.set_debug(0)
    assert st["instance_path"] == 1 and r_st["instance_path"] == 0
    assert st["rays"] == r_st["rays"]
    assert np.array_equal(hits, r_hits), f"{int((hits != r_hits).sum())} hit records differ"
    assert np.array_equal(rgb, r_rgb)


def test_camera_around_david_sequence_sample(tmp_path):
    """Twelve frames spread over the 360-frame sequence the benchmark times, each through the
    animation pipeline (rtx_host_load_next keeps the geometry, lights updated, runner.py:105-141
    renders them one process each) at the full 800x800, against the oracle: hit records exact,
    picture to the bar; three of them in flight at a time on one device scene."""
    import os
    threads = min(16, os.cpu_count() or 1)
    frames = [30 * k + 7 for k in range(12)]
    files = {f: bench.make_workload("camera_around_david", bench.scene_cache_dir(), f)[0] for f in frames}
    hs = rt.HostScene(files[frames[0]])
    ds = rt.DeviceScene(hs.desc, 0)
    slots = [ds.frame() for _ in range(3)]
    cams, got, q = {}, {}, []
    for i, f in enumerate(frames):
        fr = slots[i % 3]
        if len(q) == 3:
            g, old = q.pop(0)
            got[g] = old.wait()
        if i > 0:
            assert hs.load_next(files[f]), "geometry kept from frame to frame"
            ds.update_lights(hs.lights())
        cams[f], _ = hs.camera(0)
        assert (cams[f].width, cams[f].height) == (800, 800)
        fr.submit(cams[f], want_hits=True)
        q.append((f, fr))
    for g, fr in q:
        got[g] = fr.wait()
    assert len({bytes(cams[f].position) for f in frames}) == len(frames)      # twelve different views
    for f in frames:
        rgb, hits, st = got[f]
        o_rgb, o_hits, o_ctr, _ = ck.oracle_render(hs.desc, cams[f], nthreads=threads)
        assert len(ck.compare_hits(hits, o_hits, allow_src_unknown=False)) == 0, f
        same, mx = ck.image_diff(rgb, o_rgb)
        assert mx <= MAX_DELTA and same >= MIN_SAME, (f, same, mx)
        assert st["rays"] == o_ctr["rays"]
    ds.close(); hs.close()


@pytest.mark.parametrize("name", ["dragon_metal", "camera_around_david", "metal_glass_plates"])
def test_wide_blas_at_size(workloads, name):
    """The eight-wide float-box records (rtx_scene_set_wide_blas) on the full-size frame: the bytes and
    hit records of the default path, from fewer records, without a redo."""
    hs, ds = workloads(name)
    cam, _ = hs.camera(0)
    rgb, hits, st = ds.render(cam, want_hits=True)
    wide = rt.DeviceScene(hs.desc, 0, wide_blas=True)          # (a second device scene: the shared one holds no such records)
    wide.set_wide_blas(True)
    try:
        w_rgb, w_hits, w_st = wide.render(cam, want_hits=True)
    finally:
        wide.close()
    assert w_st["wide_blas"] == 1 and w_st["wide_blas_redone"] == 0
    assert np.array_equal(hits, w_hits), f"{int((hits != w_hits).sum())} hit records differ"
    assert np.array_equal(rgb, w_rgb)
    assert w_st["rays"] == st["rays"]
    if name != "metal_glass_plates":          # (its meshes are a few triangles each: nothing to save)
        assert w_st["record_fetches"] < 0.9 * st["record_fetches"]
