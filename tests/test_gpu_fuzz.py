"""Random small scenes through the HIP path (C-ABI) and through the oracle: meshes of random triangles under random
rotations, scales and translations (BVHs from rt_mesh_bvh_build, both split rules, several leaf sizes), spheres, all
three materials, several lights, reflection depths 0..3, and frames of odd size, whose centre ray runs exactly along
the z axis.  Hit keys must be the oracle's, all of them; NaN pixels must be NaN in both; the other pixel values may
differ by the device math library's powf / expf against the host's (gamma, specular exponent), which stays under 1e-6
relative - the task's tolerance is 1e-4."""
import ctypes as C

import numpy as np
import pytest

import hand_scenes as hs
from cudaraytracer_amd import _native as N
from parity import compare

pytestmark = pytest.mark.gpu


def unit(v):
    return (v / np.linalg.norm(v)).tolist()


def random_triangles(rng, n):
    tris = []
    for t in range(n):
        c = rng.uniform(-2.0, 2.0, 3)
        p = c + rng.uniform(-1.5, 1.5, (3, 3))
        if np.cross(p[1] - p[0], p[2] - p[0])[2] > 0 and t % 3:  # two in three face the camera of an unrotated mesh
            p = p[[0, 2, 1]]
        if t % 5 == 0:  # a triangle in a plane z = const, x or y aligned: rays along the axes meet box faces
            p[:, 2] = np.round(c[2] * 4.0) / 4.0
            p[1, 1] = p[0, 1]
        uv = rng.uniform(0.0, 1.0, (3, 2))
        nrm = unit(rng.normal(0, 1, 3))
        tris.append([{"pos": p[k].astype(np.float32).tolist(), "uv": uv[k].tolist(), "normal": nrm} for k in range(3)])
    return tris


def bvh_of(rtlib, tris, rule, threshold, tight):
    """(triangles in leaf order, node tuples for hand_scenes.Mesh) from the library's host builder."""
    n = len(tris)
    verts = (N.RtVertex * (3 * n))()
    idx = (N.RtTriangle * n)()
    for t, corners in enumerate(tris):
        for k, c in enumerate(corners):
            verts[3 * t + k].position[:] = c["pos"]
            idx[t].i[k] = 3 * t + k
    nodes = (N.RtLinearNode * (2 * n))()
    cnt = C.c_int()
    order = (C.c_int * n)()
    rc = rtlib.rt_mesh_bvh_build(C.addressof(verts), 3 * n, C.addressof(idx), n, rule, threshold, tight, C.addressof(nodes), 2 * n, C.byref(cnt),
                                 C.addressof(order))
    assert rc == 0, rtlib.rt_last_error(None)
    spec = [(list(nd.bbmin), list(nd.bbmax), nd.offset, nd.primCnt, nd.axis) for nd in nodes[: cnt.value]]
    return [tris[i] for i in order], spec


def random_transform(rng, kind):
    """Object -> world as a row-vector matrix (p_world = p_obj . M) and its inverse."""
    angle = {0: 0.0, 1: np.pi / 2, 2: np.pi / 4}.get(kind, rng.uniform(0, 2 * np.pi))
    axis = np.array([0.0, 0.0, 1.0]) if kind in (0, 1, 2) else np.array(unit(rng.normal(0, 1, 3)))
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    rot = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                    [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                    [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    scale = 1.0 if kind == 0 else rng.uniform(0.5, 1.8)
    m = np.eye(4)
    m[:3, :3] = rot.T * scale
    m[3, :3] = [rng.uniform(-2.0, 2.0), rng.uniform(-1.5, 1.5), rng.uniform(4.0, 9.0)]
    return m, np.linalg.inv(m)


def random_scene(rtlib, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(8, 25)) * 2 + 1, int(rng.integers(8, 19)) * 2 + 1
    meshes = []
    for mi in range(int(rng.integers(1, 4))):
        tris = random_triangles(rng, int(rng.integers(1, 120)))
        tris, spec = bvh_of(rtlib, tris, int(rng.integers(0, 2)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 2)))
        m, inv = random_transform(rng, int(rng.integers(0, 6)))
        pos = np.array([c["pos"] for t in tris for c in t], dtype=np.float64)
        world = np.concatenate([pos, np.ones((len(pos), 1))], axis=1) @ m
        box = ((world[:, :3].min(0) - 1e-3).tolist(), (world[:, :3].max(0) + 1e-3).tolist())
        tex = rng.integers(0, 256, (2, 2, 3)).tolist()
        meshes.append(hs.Mesh(tris, material=int(rng.choice([hs.DIFFUSE, hs.DIFFUSE, hs.MIRROR, hs.GLASS])), ior=float(rng.uniform(1.1, 1.7)),
                              ks=float(rng.uniform(0, 0.5)), diffuse=tex, specular=rng.integers(0, 256, (1, 1, 3)).tolist(), nodes=spec,
                              mode=int(rng.integers(0, 2)), transform_inverse=inv.astype(np.float32), world_box=box))
    if seed % 4 != 3:  # a wall behind everything, so that most pixels are shaded and most shadow rays have something to meet
        wall = [hs.tri((-30, -30, 14), (-30, 60, 14), (60, -30, 14)), hs.tri((40, 40, 15), (40, -60, 15), (-60, 40, 15))]
        meshes.append(hs.Mesh(wall, diffuse=rng.integers(0, 256, (2, 2, 3)).tolist(), ks=0.2, specular=[[(200, 200, 200)]]))
    spheres = []
    for _ in range(int(rng.integers(0, 5))):
        centre = [float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2)), float(rng.uniform(3, 10))]
        spheres.append((centre, float(rng.uniform(0.3, 1.4)), int(rng.choice([hs.DIFFUSE, hs.MIRROR, hs.GLASS])), float(rng.uniform(1.1, 1.7)),
                        rng.uniform(0.1, 1.0, 3).tolist()))
    lights = [((0, 0, 0), (1, 1, 1), 5.0)] if seed % 3 == 0 else []
    for _ in range(int(rng.integers(1, 4))):
        lights.append((rng.uniform(-6, 6, 3).tolist(), rng.uniform(0.2, 1.0, 3).tolist(), float(rng.uniform(2, 30))))
    return hs.Scene(w, h, meshes, spheres, lights, max_depth=int(rng.integers(0, 4)), bias=1e-3, gamma=int(rng.integers(0, 2)))


@pytest.mark.parametrize("seed", range(96))
def test_random_scene_matches_oracle(gpu_ctx, oracle, rtlib, seed):
    scene = random_scene(rtlib, seed)
    ref_rgba, ref_keys = scene.render_oracle(oracle)
    gpu_ctx.set_path_budget(16)  # a glass surface has two children: up to 1 + 2 + 4 + 8 path nodes per pixel at depth 3
    try:
        rgba, keys = scene.render_gpu(gpu_ctx)
    finally:
        gpu_ctx.set_path_budget(3)
    hit = int((ref_keys[..., 0] >= 0).sum())
    assert hit > 0, "the random scene shows nothing: the generator needs another look"
    assert np.array_equal(keys, ref_keys), f"{int((keys != ref_keys).any(axis=2).sum())} of {keys.shape[0] * keys.shape[1]} hit keys differ"
    assert np.array_equal(np.isnan(rgba), np.isnan(ref_rgba))
    r = compare(rgba, keys, ref_rgba, ref_keys)
    assert r["max_rel"] <= 1e-6 and r["radiance_ok_fraction"] == 1.0, r
