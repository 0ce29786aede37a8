"""rt_set_upload_refit: node arrays as the reference builds them (FLT_MIN-clamped boxes, cudaTracer/BoundingBox.cuh:25-27,
cudaTracer/CudaMesh.h:58-63) go in, the library recomputes the boxes on the device, and the frame is the oracle's frame
on the same tree with the host refit's boxes - hit keys, radiance and integer-equal test counters."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity
from test_bvh_refit import refit_scene_in_place

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def refit_ctx():
    ctx = rt.Context(0)
    ctx.set_box_test_split(0)
    ctx.set_upload_refit(True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name,w,h,size,lights,depth", [("soup", 160, 96, 50000, 2, 0), ("mixed", 160, 120, 0, 4, 3), ("cubes", 160, 120, 8, 3, 2),
                                                         ("heightfield", 160, 96, 20000, 2, 0)])
def test_reference_nodes_refit_at_upload(refit_ctx, gpu_ctx, oracle, rtlib, synthetic_assets, name, w, h, size, lights, depth):
    scene = rt.HostScene(name, w, h, size, lights, depth, asset_dir=synthetic_assets if name == "mixed" else None, tight_bvh=False)
    # the reference's boxes as they come: the frame and what it costs
    loose_rgba, loose_keys, loose_cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    # the same arrays through a context that refits at upload; the packed blob (host refit) against the device's
    dscene = refit_ctx.upload(scene)
    ptr, n = refit_ctx.scene_blob(dscene)
    dev_blob = refit_ctx.download(ptr, (n,), np.uint8)
    host_blob = refit_ctx.pack(scene.desc)
    differing = np.flatnonzero(dev_blob != host_blob)
    assert differing.size == 0, f"device refit and host refit disagree in {differing.size} bytes, first at offset {differing[:8]}"
    assert not np.array_equal(host_blob, gpu_ctx.pack(scene.desc))  # and the refit did change boxes
    rgba, keys, cnt = refit_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
    plain = refit_ctx.render_frame(scene, dscene, want_keys=True)  # the walk over quads, built from the refitted pairs
    refit_ctx.free_scene(dscene)
    # the oracle on the host-refitted arrays
    refit_scene_in_place(rtlib, scene.desc)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt, (cnt, ref_cnt)
    assert np.array_equal(plain[1], keys) and np.array_equal(plain[0], rgba, equal_nan=True)
    # same image as on the reference's boxes, for fewer tests
    assert np.array_equal(keys, loose_keys) and np.array_equal(rgba, loose_rgba, equal_nan=True)
    assert cnt["node"] < loose_cnt["node"] and cnt["tri"] <= loose_cnt["tri"]


def test_refit_at_full_size(refit_ctx, gpu_ctx, oracle, rtlib):
    """The headline's triangles as the reference would hand them over: 1 M-triangle soup, median-cut tree with
    FLT_MIN-clamped boxes, 1080p.  With the refit the frame costs what a tight tree costs, and equals the oracle's."""
    w, h = 1920, 1080
    scene = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=False)
    dscene = refit_ctx.upload(scene)
    try:
        rgba, keys, cnt = refit_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
    finally:
        refit_ctx.free_scene(dscene)
    scene_loose = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=False)
    refit_scene_in_place(rtlib, scene.desc)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=16, shadow_any_hit=True)
    r = assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert r["key_mismatches"] == 0 and cnt == ref_cnt, (r, cnt, ref_cnt)
    # the same arrays without the refit: the same picture for several times the tests (about 2 100 box and 6 800 triangle
    # tests per ray; the refitted tree keeps the reference's cuts, which were made on clamped centroids, and stays
    # behind a tree built on true bounds from the start)
    loose_rgba, loose_keys, loose_cnt = gpu_ctx.render_frame(scene_loose, want_keys=True, want_counters=True)
    assert np.array_equal(loose_keys, keys) and np.array_equal(loose_rgba, rgba, equal_nan=True)
    print("tests per ray, reference boxes / refitted:", {k: (loose_cnt[k] / (cnt["primary"] + cnt["shadow"]), cnt[k] / (cnt["primary"] + cnt["shadow"])) for k in ("node", "tri")})
    assert 3 * cnt["node"] < loose_cnt["node"] and 5 * cnt["tri"] < loose_cnt["tri"]
    # the tree built with true bounds from the start shows the same picture
    tight = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=True)
    t_rgba, t_keys, _ = gpu_ctx.render_frame(tight, want_keys=True)
    assert np.array_equal(t_keys, keys) and np.array_equal(t_rgba, rgba, equal_nan=True)


def test_cpp_driver_refits_by_default(oracle):
    """RayTracer.h uploads with rt_set_upload_refit on: reference-shaped trees, true boxes, the oracle's picture."""
    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 3, 2)
    rgba = scene.run_frames(1, 0.0)
    ref_rgba, _, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8)
    assert_parity(rgba, None, ref_rgba, None)
