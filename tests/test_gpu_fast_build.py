"""The FAST build of the kernels (rt_set_arithmetic: fused multiply-adds, approximate division / square root, min / max
box test, fp32 texture weights) against the CPU oracle, which is the EXACT arithmetic.

Bar (the tolerance this path is held to): hit keys agree on >= 99.99 % of the pixels; of the pixels whose hit keys
agree, radiance is within 1e-4 relative or one step of the 8-bit output on >= 99.9 % (a shadow ray or a reflection that
grazes an edge can flip for a pixel whose primary hit is the same).  Pixels that differ are listed and must be
explainable as ties: the two hit distances differ by rounding only.
"""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import KEY_AGREEMENT, compare

pytestmark = pytest.mark.gpu

# the five workload shapes: cube grid (cfg 1), triangle soup (cfg 2), OBJ barrel scene (cfg 3), height field (cfg 4),
# default-scene analogue with 16 lights and depth 4 (cfg 5)
CASES = [
    ("cubes", 640, 480, 16, 4, 5, False),
    ("soup", 480, 270, 200000, 1, 0, True),
    ("soup", 480, 270, 200000, 1, 0, False),
    ("barrel", 480, 270, 0, 10, 2, True),
    ("heightfield", 480, 270, 200000, 1, 0, True),
    ("mixed", 480, 270, 0, 16, 4, True),
]


@pytest.fixture()
def fast_ctx():
    ctx = rt.Context(0)
    ctx.set_arithmetic(True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name,w,h,size,lights,depth,tight", CASES)
def test_fast_build_within_tolerance_of_the_oracle(fast_ctx, oracle, synthetic_assets, name, w, h, size, lights, depth, tight):
    scene = rt.HostScene(name, w, h, size, lights, depth, asset_dir=synthetic_assets if name in ("barrel", "mixed") else None, tight_bvh=tight)
    ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, _ = fast_ctx.render_frame(scene, want_keys=True)
    same_key = (keys == ref_keys).all(-1)
    agreement = float(same_key.mean())
    mism = np.argwhere(~same_key)
    for (y, x) in mism[:10]:
        print(f"{name}: pixel ({x},{y}) fast key {keys[y, x].tolist()} oracle key {ref_keys[y, x].tolist()}")
    assert agreement >= KEY_AGREEMENT, (name, agreement, len(mism))
    r = compare(rgba[same_key][None], None, ref_rgba[same_key][None], None)
    print(name, "key agreement", agreement, r)
    assert r["radiance_ok_fraction"] >= 0.999, r


def test_fast_and_exact_builds_share_a_context(gpu_ctx, oracle):
    """Switching back restores bit-identical frames; the fast frame stays within tolerance."""
    w, h = 160, 120
    scene = rt.HostScene("mixed", w, h, 0, 4, 3)
    exact, keys, _ = gpu_ctx.render_frame(scene, want_keys=True)
    gpu_ctx.set_arithmetic(True)
    try:
        fast, fkeys, _ = gpu_ctx.render_frame(scene, want_keys=True)
    finally:
        gpu_ctx.set_arithmetic(False)
    again, keys2, _ = gpu_ctx.render_frame(scene, want_keys=True)
    assert np.array_equal(exact, again, equal_nan=True) and np.array_equal(keys, keys2)
    assert float((keys == fkeys).all(-1).mean()) >= KEY_AGREEMENT
