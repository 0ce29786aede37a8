"""Known answers derived by hand - not by the oracle - for the parts of the path the reference's own unit tests do not
cover: Moller-Trumbore, the sphere test, Fresnel, refraction, the bilinear filter, the equal-distance tie rule and the
near / far visiting order.  Each value follows from the cited reference lines with pencil and paper
(tests/hand_scenes.py states the derivations); the CPU oracle must reproduce them here, the HIP path in
tests/test_gpu_known_answers.py."""
import ctypes as C

import numpy as np
import pytest

import hand_scenes as hs

f3 = lambda *v: (C.c_float * 3)(*v)  # noqa: E731
BIG = float(np.finfo(np.float32).max)


def test_triangle_through_the_centroid(oracle):
    """cudaTracer/Mesh.cuh:182-212: e1 = (0, 9, 0), e2 = (9, 0, 0), p = d x e2 = (0, 9, 0), det = 81; from (0, 0, 0) along +z:
    u = v = 27 / 81 = 1/3, t = 405 / 81 = 5."""
    lib = oracle.load()
    out = f3(0, 0, 0)
    v0, v1, v2 = (f3(*p) for p in hs.FRONT)
    assert lib.oracle_triangle_intersect(v0, v1, v2, f3(0, 0, 0), f3(0, 0, 1), BIG, out) == 1
    assert out[0] == 5.0 and abs(out[1] - 1 / 3) < 1e-6 and abs(out[2] - 1 / 3) < 1e-6
    # wound the other way the determinant is negative: culled
    assert lib.oracle_triangle_intersect(v0, v2, v1, f3(0, 0, 0), f3(0, 0, 1), BIG, out) == 0
    # a hit exactly at the ray's reach is accepted (t <= tMax), one beyond it is not; behind the origin never
    assert lib.oracle_triangle_intersect(v0, v1, v2, f3(0, 0, 0), f3(0, 0, 1), 5.0, out) == 1
    assert lib.oracle_triangle_intersect(v0, v1, v2, f3(0, 0, 0), f3(0, 0, 1), float(np.nextafter(np.float32(5), np.float32(0))), out) == 0
    assert lib.oracle_triangle_intersect(v0, v1, v2, f3(0, 0, 6), f3(0, 0, 1), BIG, out) == 0
    # through a corner region outside the triangle: u + v > det
    assert lib.oracle_triangle_intersect(v0, v1, v2, f3(5, 5, 0), f3(0, 0, 1), BIG, out) == 0


def test_sphere_on_the_axis(oracle):
    """cudaTracer/Sphere.cuh:9-54 with c = (0, 0, 5), r = 1: from the origin a = 1, b = -10, c = 24, roots 4 and 6; from the
    centre the roots are -1 and 1; from z = 7 both are negative."""
    lib = oracle.load()
    t = C.c_float(0)
    assert lib.oracle_sphere_intersect(f3(0, 0, 5), 1.0, f3(0, 0, 0), f3(0, 0, 1), BIG, C.byref(t)) == 1 and t.value == 4.0
    assert lib.oracle_sphere_intersect(f3(0, 0, 5), 1.0, f3(0, 0, 5), f3(0, 0, 1), BIG, C.byref(t)) == 1 and t.value == 1.0
    assert lib.oracle_sphere_intersect(f3(0, 0, 5), 1.0, f3(0, 0, 7), f3(0, 0, 1), BIG, C.byref(t)) == 0
    assert lib.oracle_sphere_intersect(f3(0, 0, 5), 1.0, f3(0, 0, 0), f3(0, 0, 1), 3.5, C.byref(t)) == 0  # beyond the reach
    assert lib.oracle_sphere_intersect(f3(0, 0, 5), 1.0, f3(0, 2, 0), f3(0, 0, 1), BIG, C.byref(t)) == 0  # passes beside it


def test_fresnel_and_refraction(oracle):
    """cudaTracer/tracer.cu:67-115.  Normal incidence: kr = ((n - 1) / (n + 1))^2 and the refracted direction is the
    incident one; beyond the critical angle from inside: kr = 1 and refract returns the zero vector."""
    lib = oracle.load()
    for n in (1.5, 1.3, 4.0):
        assert abs(lib.oracle_fresnel(f3(0, 0, 1), f3(0, 0, -1), n) - ((n - 1) / (n + 1)) ** 2) < 1e-6
    grazing_inside = f3(float(np.sqrt(0.75)), 0, 0.5)  # cos = 0.5 against a normal along the ray: 1.5 * sin 60 > 1
    assert lib.oracle_fresnel(grazing_inside, f3(0, 0, 1), 1.5) == 1.0
    out = f3(9, 9, 9)
    lib.oracle_refract(f3(0, 0, 1), f3(0, 0, -1), 1.5, out)
    assert np.allclose(list(out), [0, 0, 1], atol=1e-6)
    lib.oracle_refract(grazing_inside, f3(0, 0, 1), 1.5, out)
    assert list(out) == [0.0, 0.0, 0.0]
    # Snell at 30 degrees into n = 1.5: sin t = 1/3
    lib.oracle_refract(f3(0.5, 0, float(np.sqrt(0.75))), f3(0, 0, -1), 1.5, out)
    assert abs(out[0] / float(np.linalg.norm(list(out))) - 1 / 3) < 1e-6


@pytest.mark.parametrize("uvm,want", hs.BILINEAR)
def test_bilinear_filter(oracle, uvm, want):
    """cudaTracer/Texture.cuh:33-69 on a 2x2 map: a texel centre returns the texel, a point midway between centres their
    mean, CLAMP repeats the edge.  (WRAP to the left of texel 0 reads out of bounds in the reference; here it wraps.)"""
    lib = oracle.load()
    tex, keep = hs.texture([[(255, 0, 0), (0, 255, 0)], [(0, 0, 255), (255, 255, 255)]], uvm[2])
    out = f3(0, 0, 0)
    lib.oracle_tex_sample(C.byref(tex), uvm[0], uvm[1], out)
    assert np.allclose(list(out), want, atol=1e-6), (uvm, list(out))


@pytest.mark.parametrize("name", sorted(hs.probes()))
def test_one_ray_scenes(oracle, name):
    scene, rgb, key = hs.probes()[name]
    rgba, keys = scene.render_oracle(oracle)
    if key is not None:
        assert tuple(keys[0, 0]) == key, (name, keys[0, 0])
    if rgb is not None:
        assert np.allclose(rgba[0, 0, :3], rgb, rtol=hs.RTOL.get(name, 2e-6), atol=1e-7), (name, rgba[0, 0])
    assert rgba[0, 0, 3] == 1.0


@pytest.mark.parametrize("uvm,want", hs.BILINEAR)
def test_bilinear_filter_through_a_frame(oracle, uvm, want):
    rgba, _ = hs.bilinear_scene(*uvm).render_oracle(oracle)
    assert np.allclose(rgba[0, 0, :3], want, atol=2e-6), (uvm, rgba[0, 0])


def test_near_far_order_flips_with_the_ray_sign(oracle):
    scene = hs.flip_scene()
    _, keys = scene.render_oracle(oracle)
    half = scene.width // 2
    assert (keys[:, :half] == np.array([0, 0, 1, 2])).all(), keys[:, :, 1]
    assert (keys[:, half:] == np.array([0, 3, 4, 5])).all(), keys[:, :, 1]


def test_pixel_to_ray_mapping(oracle):
    scene, (x, y), rgb = hs.camera_scene()
    rgba, keys = scene.render_oracle(oracle)
    seen = keys[:, :, 0] >= 0
    assert seen[y, x] and seen.sum() == 1, keys[:, :, 0]
    assert np.allclose(rgba[y, x, :3], rgb, rtol=1e-5), rgba[y, x]
    others = rgba[~seen][:, :3]
    assert np.allclose(others, hs.BG, atol=1e-7), others
