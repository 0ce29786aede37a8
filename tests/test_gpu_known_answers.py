"""The hand-derived known answers of tests/test_known_answers.py through the HIP path (C-ABI): one tiny scene per
probe, compared with the value derived on paper, not with the oracle."""
import numpy as np
import pytest

import hand_scenes as hs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(hs.probes()))
def test_one_ray_scenes(gpu_ctx, name):
    scene, rgb, key = hs.probes()[name]
    rgba, keys = scene.render_gpu(gpu_ctx)
    if key is not None:
        assert tuple(keys[0, 0]) == key, (name, keys[0, 0])
    if rgb is not None:
        assert np.allclose(rgba[0, 0, :3], rgb, rtol=hs.RTOL.get(name, 2e-6), atol=1e-7), (name, rgba[0, 0])
    assert rgba[0, 0, 3] == 1.0


@pytest.mark.parametrize("uvm,want", hs.BILINEAR)
def test_bilinear_filter(gpu_ctx, uvm, want):
    rgba, _ = hs.bilinear_scene(*uvm).render_gpu(gpu_ctx)
    assert np.allclose(rgba[0, 0, :3], want, atol=2e-6), (uvm, rgba[0, 0])


def test_pixel_to_ray_mapping(gpu_ctx):
    scene, (x, y), rgb = hs.camera_scene()
    rgba, keys = scene.render_gpu(gpu_ctx)
    seen = keys[:, :, 0] >= 0
    assert seen[y, x] and seen.sum() == 1, keys[:, :, 0]
    assert np.allclose(rgba[y, x, :3], rgb, rtol=1e-5), rgba[y, x]
    assert np.allclose(rgba[~seen][:, :3], hs.BG, atol=1e-7)


def test_near_far_order_flips_with_the_ray_sign(gpu_ctx):
    scene = hs.flip_scene()
    _, keys = scene.render_gpu(gpu_ctx)
    half = scene.width // 2
    assert (keys[:, :half] == np.array([0, 0, 1, 2])).all(), keys[:, :, 1]
    assert (keys[:, half:] == np.array([0, 3, 4, 5])).all(), keys[:, :, 1]


def test_ray_along_an_axis_plane_of_a_rotated_mesh(gpu_ctx, oracle):
    """World-space directions without a zero component, but in the space of a mesh turned 45 degrees about z the
    rays of the frame's diagonal (d.x == d.y) run along the plane x = 0: their reciprocal direction gets infinite at
    the mesh's entry, in the middle of the walk.  Every pixel must equal the oracle's, bit for bit."""
    c = float(np.float32(np.sqrt(0.5)))
    inv = [[c, c, 0, 0], [-c, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]  # row vectors: d' = d . M, so d'.x = (d.x - d.y) * c
    big = ((-30, -30, 5), (-30, 60, 5), (60, -30, 5))
    scene = hs.Scene(4, 4, [hs.Mesh([hs.tri(*big)], transform_inverse=inv, world_box=((-100, -100, 0), (100, 100, 10)))],
                     lights=[((0, 0, 0), (1, 1, 1), 5.0), ((3, 3, 0), (1, 0.5, 0.25), 2.0)])
    ref_rgba, ref_keys = scene.render_oracle(oracle)
    rgba, keys = scene.render_gpu(gpu_ctx)
    assert (ref_keys[..., 0] == 0).sum() >= 12  # the triangle fills the frame but for a corner
    assert np.array_equal(keys, ref_keys)
    assert np.array_equal(rgba, ref_rgba, equal_nan=True)


def test_ray_with_a_nan_in_its_origin_misses_like_the_reference(gpu_ctx, oracle):
    """A NaN in one coordinate of the origin makes that axis' plane distances NaN.  The reference's chain of comparisons
    (cudaTracer/BoundingBox.cuh:115-139) then never passes a box; a min / max form would drop the NaN operands and test the
    other two axes.  Such rays must go the literal way: every pixel as the oracle has it (the background)."""
    big = ((-30, -30, 5), (-30, 60, 5), (60, -30, 5))
    scene = hs.Scene(9, 7, [hs.Mesh([hs.tri(*big)])], spheres=[((0, 0, 8), 2.0, hs.DIFFUSE, 1.3, (1, 1, 1))], lights=hs.LIGHT_AT_CAMERA,
                     origin=(float("nan"), 0.0, 0.0))
    ref_rgba, ref_keys = scene.render_oracle(oracle)
    rgba, keys = scene.render_gpu(gpu_ctx)
    assert (ref_keys[..., 0] == -1).all()  # what the reference's arithmetic makes of it: nothing is hit
    assert np.array_equal(keys, ref_keys)
    assert np.array_equal(rgba, ref_rgba, equal_nan=True)
