"""rt_set_unlit_shadow_skip through the C-ABI: frames with the switch on equal frames with it off bit for bit - pixels,
hit keys, counters - on every workload shape, and the shadow launch walks fewer rays."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
import hand_scenes as hs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,w,h,size,lights,depth", [
    ("soup", 480, 270, 200000, 1, 0),
    ("soup", 257, 131, 50000, 5, 0),
    ("cubes", 320, 240, 8, 4, 3),
    ("heightfield", 320, 180, 200000, 3, 0),
    ("mixed", 320, 240, 0, 6, 3),
    ("barrel", 320, 240, 0, 4, 2),
])
def test_frames_do_not_depend_on_the_switch(gpu_ctx, synthetic_assets, name, w, h, size, lights, depth):
    scene = rt.HostScene(name, w, h, size, lights, depth, asset_dir=synthetic_assets if name in ("mixed", "barrel") else None, tight_bvh=True)
    dscene = gpu_ctx.upload(scene)
    try:
        plain = gpu_ctx.render_frame(scene, dscene, want_keys=True)
        counted = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        gpu_ctx.set_unlit_shadow_skip(True)
        try:
            skipped = gpu_ctx.render_frame(scene, dscene, want_keys=True)
            skipped_counted = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        finally:
            gpu_ctx.set_unlit_shadow_skip(False)
        assert np.array_equal(skipped[1], plain[1])
        assert np.array_equal(skipped[0], plain[0], equal_nan=True)
        assert np.array_equal(skipped_counted[0], counted[0], equal_nan=True) and skipped_counted[2] == counted[2]
        assert (plain[1][..., 0] >= 0).any()
    finally:
        gpu_ctx.free_scene(dscene)


def test_light_behind_the_surface_and_the_cases_that_are_walked(gpu_ctx, oracle):
    """One-ray scenes: a light behind a one-sided surface (lambert 0: skipped, pixel black either way); a light AT the surface
    point (distance 0: the reference's term is a NaN, the ray is walked); a light of enormous power behind the surface
    (walked).  With the switch on, every pixel equals the oracle's, NaN for NaN."""
    behind = [((0, 0, 10), (1, 1, 1), 5.0)]
    at_point = [((0, 0, 5), (1, 1, 1), 5.0)]
    huge = [((0, 0, 10), (1e30, 1e30, 1e30), 1e30)]
    gpu_ctx.set_unlit_shadow_skip(True)
    try:
        for lights in (behind, at_point, huge, behind + hs.LIGHT_AT_CAMERA + at_point):
            scene = hs.Scene(1, 1, [hs.Mesh([hs.tri(*hs.FRONT)])], lights=lights)
            ref_rgba, ref_keys = scene.render_oracle(oracle)
            rgba, keys = scene.render_gpu(gpu_ctx)
            assert np.array_equal(keys, ref_keys)
            assert np.array_equal(rgba, ref_rgba, equal_nan=True), (lights, rgba, ref_rgba)
        # a sphere whose Ks is infinite, lit from behind and from the front: 0 * inf is a NaN the reference lets through
        # only for a visible light
        for lights in (behind, behind + hs.LIGHT_AT_CAMERA):
            scene = hs.Scene(1, 1, spheres=[((0, 0, 5), 1.0, hs.DIFFUSE, 1.3, (0.4, 0.2, 0.1))], lights=lights)
            scene.shapes[0].material.Ks = float("inf")
            ref_rgba, ref_keys = scene.render_oracle(oracle)
            rgba, keys = scene.render_gpu(gpu_ctx)
            assert np.array_equal(keys, ref_keys)
            assert np.array_equal(rgba, ref_rgba, equal_nan=True), (lights, rgba, ref_rgba)
    finally:
        gpu_ctx.set_unlit_shadow_skip(False)
