"""rt_set_upload_rebuild: node arrays as the reference builds them go in (FLT_MIN-clamped boxes AND cuts made on
FLT_MIN-clamped centroid bounds, cudaTracer/BVH.h:122-128), the library ignores them and builds every mesh BVH on the
device from the triangles.  Same picture and hit keys as the reference's own tree gives, for the tests of a tree cut
on true bounds."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rebuild_ctx():
    ctx = rt.Context(0)
    ctx.set_upload_rebuild(True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name,w,h,size,lights,depth", [("soup", 160, 96, 50000, 2, 0), ("mixed", 160, 120, 0, 4, 3), ("heightfield", 160, 96, 20000, 2, 0)])
def test_reference_nodes_rebuilt_at_upload(rebuild_ctx, gpu_ctx, oracle, synthetic_assets, name, w, h, size, lights, depth):
    assets = synthetic_assets if name == "mixed" else None
    scene = rt.HostScene(name, w, h, size, lights, depth, asset_dir=assets, tight_bvh=False)
    loose_rgba, loose_keys, loose_cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    rgba, keys, cnt = rebuild_ctx.render_frame(scene, want_keys=True, want_counters=True)
    plain = rebuild_ctx.render_frame(scene, want_keys=True)
    assert np.array_equal(plain[1], keys) and np.array_equal(plain[0], rgba, equal_nan=True)
    # the reference's picture (the oracle on the reference's own arrays), whatever the tree
    ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert np.array_equal(keys, loose_keys)
    # ... for the tests of a tree built on true bounds: the same scene with such a tree from the host builder
    tight = rt.HostScene(name, w, h, size, lights, depth, asset_dir=assets, tight_bvh=True)
    _, t_keys, t_cnt = gpu_ctx.render_frame(tight, want_keys=True, want_counters=True)
    assert np.array_equal(t_keys, keys)
    rays = cnt["primary"] + cnt["shadow"] + cnt["secondary"]
    print(name, "box tests per ray: reference arrays", loose_cnt["node"] / rays, "rebuilt", cnt["node"] / rays, "host tree on true bounds", t_cnt["node"] / rays)
    assert cnt["node"] < loose_cnt["node"] and cnt["tri"] <= loose_cnt["tri"]
    assert cnt["node"] <= 1.05 * t_cnt["node"]


def test_rebuild_at_full_size(rebuild_ctx, gpu_ctx):
    """The headline's triangles as the reference would hand them over: rebuilt at upload, the 1080p frame costs what the
    median-cut tree on true bounds costs (about 97 box + 80 triangle tests per ray instead of 2 100 + 6 800) and shows
    the same picture."""
    w, h = 1920, 1080
    scene = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=False)
    rgba, keys, cnt = rebuild_ctx.render_frame(scene, want_keys=True, want_counters=True)
    tight = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=True)
    t_rgba, t_keys, t_cnt = gpu_ctx.render_frame(tight, want_keys=True, want_counters=True)
    assert np.array_equal(t_keys, keys) and np.array_equal(t_rgba, rgba, equal_nan=True)
    rays = cnt["primary"] + cnt["shadow"]
    print("tests per ray, rebuilt:", cnt["node"] / rays, cnt["tri"] / rays, "host tree on true bounds:", t_cnt["node"] / rays, t_cnt["tri"] / rays)
    assert cnt["node"] <= 1.02 * t_cnt["node"] and cnt["tri"] <= 1.02 * t_cnt["tri"]
