"""The CPU oracle against the committed golden vectors (tests/golden/, written by
tests/golden/make_golden.py).  The GPU counterpart lives in test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

import cudaraytracer_amd as rt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402


@pytest.mark.parametrize("key", sorted(make_golden.FRAMES))
def test_oracle_reproduces_golden_frame(oracle, key):
    name, w, h, size, lights, depth, tight = make_golden.FRAMES[key]
    g = np.load(os.path.join(GOLDEN, f"frame_{key}.npz"))
    s = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
    # thread count and row interleaving must not matter
    rgba, keys, cnt = oracle.render(s.desc, s.opts, s.camera, w, h, threads=3, shadow_any_hit=True)
    assert np.array_equal(keys, g["keys"])
    assert np.array_equal(rgba, g["rgba"], equal_nan=True)
    assert [cnt[n] for n in rt.RT_COUNTER_NAMES] == g["counters"].tolist()


def test_oracle_reproduces_golden_probes():
    g = np.load(os.path.join(GOLDEN, "probes.npz"))
    got = make_golden.eval_probes(g)
    for k, v in got.items():
        assert np.array_equal(v, g[k], equal_nan=True), k
    # the table exercises both outcomes of every predicate
    assert 0 < g["box_hit"].sum() < len(g["box_hit"])
    assert 0 < g["tri_out"][:, 0].sum() < len(g["tri_out"])
    assert 0 < g["sph_out"][:, 0].sum() < len(g["sph_out"])
    tir = (g["fr_refr"] == 0).all(1)
    assert 0 < tir.sum() < len(tir)
    assert np.all(g["fr_kr"][tir] == 1.0)  # total internal reflection <=> kr == 1 (cudaTracer/tracer.cu fresnel)


def test_golden_frames_have_content():
    for key in make_golden.FRAMES:
        g = np.load(os.path.join(GOLDEN, f"frame_{key}.npz"))
        hit = g["keys"][..., 0] >= 0
        assert 0.005 < hit.mean() < 1.0, key


def test_oracle_reproduces_texture_and_bvh_tables():
    g = np.load(os.path.join(GOLDEN, "tables.npz"))
    assert np.array_equal(make_golden.eval_texture(g["texels"], g["uv"], 0), g["wrap"])
    assert np.array_equal(make_golden.eval_texture(g["texels"], g["uv"], 1), g["clamp"])
    for tight, key in ((0, "ref"), (1, "tight")):
        nodes, order = make_golden.eval_bvh(g["corners"], tight)
        assert np.array_equal(nodes, g[f"nodes_{key}"]) and np.array_equal(order, g[f"order_{key}"])
    # inside the texture the two addressing modes agree; they differ only near the borders
    inner = ((g["uv"] > 0.2) & (g["uv"] < 0.8)).all(1)
    assert np.array_equal(g["wrap"][inner], g["clamp"][inner]) and not np.array_equal(g["wrap"], g["clamp"])
    # texel centres return the texel itself (cudaTracer/Texture.cuh:23-47: u * W - 0.5)
    centre = make_golden.eval_texture(g["texels"], np.float32([[0.375, 0.625]]), 1)[0]
    assert np.array_equal(centre, g["texels"][2, 1, :3].astype(np.float32) / np.float32(255.0))


def test_host_bvh_builder_reproduces_golden_nodes(rtlib):
    """The product's builder (csrc/host/BVH.h) against the committed node dump, without the oracle in the loop."""
    import ctypes as C

    from cudaraytracer_amd import _native as N

    g = np.load(os.path.join(GOLDEN, "tables.npz"))
    corners = g["corners"]
    n = len(corners)
    verts = (N.RtVertex * (3 * n))()
    tris = (N.RtTriangle * n)()
    flat = corners.reshape(-1, 3)
    for i in range(3 * n):
        verts[i].position[:] = [float(x) for x in flat[i]]
    for t in range(n):
        tris[t].i[:] = [3 * t, 3 * t + 1, 3 * t + 2]
    for tight, key in ((0, "ref"), (1, "tight")):
        nodes = (N.RtLinearNode * (2 * n))()
        cnt = C.c_int()
        order = (C.c_int * n)()
        assert rtlib.rth_bvh_triangles(verts, 3 * n, tris, n, 10, tight, nodes, C.byref(cnt), order) == 0
        assert bytes(nodes)[: 32 * cnt.value] == g[f"nodes_{key}"].tobytes()
        assert list(order) == g[f"order_{key}"].tolist()
