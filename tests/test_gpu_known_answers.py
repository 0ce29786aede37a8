"""The hand-derived known answers of tests/test_known_answers.py through the HIP path (C-ABI): one tiny scene per
probe, compared with the value derived on paper, not with the oracle."""
import numpy as np
import pytest

import hand_scenes as hs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("name", sorted(hs.probes()))
def test_one_ray_scenes(gpu_ctx, name, fast):
    scene, rgb, key = hs.probes()[name]
    gpu_ctx.set_arithmetic(fast)
    try:
        rgba, keys = scene.render_gpu(gpu_ctx)
    finally:
        gpu_ctx.set_arithmetic(False)
    if key is not None:
        assert tuple(keys[0, 0]) == key, (name, keys[0, 0])
    if rgb is not None:
        assert np.allclose(rgba[0, 0, :3], rgb, rtol=2e-6, atol=1e-7), (name, rgba[0, 0])
    assert rgba[0, 0, 3] == 1.0


@pytest.mark.parametrize("uvm,want", hs.BILINEAR)
def test_bilinear_filter(gpu_ctx, uvm, want):
    rgba, _ = hs.bilinear_scene(*uvm).render_gpu(gpu_ctx)
    assert np.allclose(rgba[0, 0, :3], want, atol=2e-6), (uvm, rgba[0, 0])


def test_near_far_order_flips_with_the_ray_sign(gpu_ctx):
    scene = hs.flip_scene()
    _, keys = scene.render_gpu(gpu_ctx)
    half = scene.width // 2
    assert (keys[:, :half] == np.array([0, 0, 1, 2])).all(), keys[:, :, 1]
    assert (keys[:, half:] == np.array([0, 3, 4, 5])).all(), keys[:, :, 1]
