"""Host BVH builder (BVH::BvhFactory) against the oracle's restatement of
cudaTracer/BVH.h, and BVH-independent properties of the traced image."""
import ctypes as C

import numpy as np
import pytest

import cudaraytracer_amd as rt
from cudaraytracer_amd import _native as N


def random_mesh(n_tris, seed, lo=-5.0, hi=5.0, coincident=0):
    rng = np.random.default_rng(seed)
    verts = (N.RtVertex * (3 * n_tris))()
    tris = (N.RtTriangle * n_tris)()
    for t in range(n_tris):
        c = rng.uniform(lo, hi, 3) if t >= coincident else np.array([1.0, 2.0, 3.0])
        for k in range(3):
            p = c + (rng.uniform(-0.5, 0.5, 3) if t >= coincident else np.array([0.1 * k, 0.0, 0.1 * (k == 2)]))
            for a in range(3):
                verts[3 * t + k].position[a] = float(np.float32(p[a]))
            tris[t].i[k] = 3 * t + k
    return verts, tris


def build_both(rtlib, oracle, verts, tris, threshold, tight):
    n = len(tris)
    out = {}
    for name in ("host", "oracle"):
        nodes = (N.RtLinearNode * (2 * n))()
        cnt = C.c_int()
        order = (C.c_int * n)()
        if name == "host":
            rc = rtlib.rth_bvh_triangles(verts, len(verts), tris, n, threshold, tight, nodes, C.byref(cnt), order)
        else:
            rc = oracle.load().oracle_bvh_triangles(C.addressof(verts), C.addressof(tris), n, threshold, tight, C.addressof(nodes), C.byref(cnt), C.addressof(order))
        assert rc == 0
        out[name] = (bytes(nodes)[: 32 * cnt.value], list(order), cnt.value)
    return out


@pytest.mark.parametrize("n,threshold,tight,lo,hi", [(1, 10, 0, -5, 5), (7, 10, 0, -5, 5), (11, 10, 0, -5, 5), (500, 10, 0, -5, 5),
                                                      (500, 10, 1, -5, 5), (2000, 4, 1, -50, 50), (333, 1, 0, 1, 9), (1000, 10, 0, -9, -1)])
def test_host_builder_equals_oracle_builder(rtlib, oracle, n, threshold, tight, lo, hi):
    verts, tris = random_mesh(n, seed=n + threshold, lo=lo, hi=hi)
    r = build_both(rtlib, oracle, verts, tris, threshold, tight)
    assert r["host"][2] == r["oracle"][2]
    assert r["host"][1] == r["oracle"][1], "leaf order differs"
    assert r["host"][0] == r["oracle"][0], "node bytes differ"
    assert sorted(r["host"][1]) == list(range(n))


def test_equal_count_split_makes_a_balanced_tree(rtlib, oracle):
    verts, tris = random_mesh(1000, seed=5)
    nodes_b, order, cnt = build_both(rtlib, oracle, verts, tris, 10, 1)["host"]
    nodes = np.frombuffer(nodes_b, dtype=np.uint8).reshape(cnt, 32)
    prim_cnt = nodes[:, 28:30].copy().view(np.uint16)[:, 0]
    leaves = prim_cnt[prim_cnt > 0]
    assert leaves.sum() == 1000 and leaves.max() <= 10
    assert cnt == 2 * len(leaves) - 1  # full binary tree
    assert len(leaves) == 128  # 1000 -> halves until <= 10: 2^7 leaves of 7-8 triangles


def test_coincident_centroids_become_one_leaf(rtlib, oracle):
    # BVH.h:126-128: no spread of the centroids along the chosen axis -> leaf, whatever the count
    verts, tris = random_mesh(40, seed=3, coincident=40)
    r = build_both(rtlib, oracle, verts, tris, 10, 0)
    assert r["host"] == r["oracle"]
    assert r["host"][2] == 1


def test_reference_bounds_are_clamped_at_flt_min(rtlib, oracle):
    """BoundingBox.cuh:25-27: the empty box's max corner is FLT_MIN, so all-negative geometry gets boxes reaching ~0."""
    verts, tris = random_mesh(64, seed=9, lo=-9, hi=-1)
    ref = np.frombuffer(build_both(rtlib, oracle, verts, tris, 10, 0)["host"][0], dtype=np.float32).reshape(-1, 8)
    tight = np.frombuffer(build_both(rtlib, oracle, verts, tris, 10, 1)["host"][0], dtype=np.float32).reshape(-1, 8)
    flt_min = np.finfo(np.float32).tiny
    assert np.all(ref[:, 3:6] == flt_min)
    assert np.all(tight[:, 3:6] < 0)


def capi_build(rtlib, verts, tris, rule, threshold, tight):
    n = len(tris)
    nodes = (N.RtLinearNode * (2 * n))()
    cnt = C.c_int()
    order = (C.c_int * n)()
    rc = rtlib.rt_mesh_bvh_build(C.addressof(verts), len(verts), C.addressof(tris), n, rule, threshold, tight, C.addressof(nodes), 2 * n,
                                 C.byref(cnt), C.addressof(order))
    assert rc == 0, rtlib.rt_last_error(None)
    return nodes, cnt.value, list(order)


def clustered_mesh(n_tris, seed):
    """Triangles in a few dense clumps of very different size, plus a thin uniform haze: far from a median cut's ideal."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-40, 40, (6, 3))
    spread = np.array([0.3, 0.5, 1.0, 2.0, 4.0, 30.0])
    verts = (N.RtVertex * (3 * n_tris))()
    tris = (N.RtTriangle * n_tris)()
    for t in range(n_tris):
        k = int(rng.integers(0, 6))
        c = centres[k] + rng.normal(0, spread[k], 3)
        for j in range(3):
            p = c + rng.uniform(-0.2, 0.2, 3)
            for a in range(3):
                verts[3 * t + j].position[a] = float(np.float32(p[a]))
            tris[t].i[j] = 3 * t + j
    return verts, tris


def walk(nodes, cnt, order, verts, tris, threshold):
    """Checks the flattened tree (first child at self + 1, second by index) and returns its surface-area cost."""
    pos = np.array([[verts[i].position[a] for a in range(3)] for i in range(len(verts))], dtype=np.float32)
    seen, cost, visited = [], 0.0, 0
    area = lambda lo, hi: float(np.prod([1.0]) * ((hi - lo)[0] * (hi - lo)[1] + (hi - lo)[1] * (hi - lo)[2] + (hi - lo)[2] * (hi - lo)[0]))
    stack = [0]
    while stack:
        i = stack.pop()
        visited += 1
        nd = nodes[i]
        lo, hi = np.array(nd.bbmin[:], dtype=np.float32), np.array(nd.bbmax[:], dtype=np.float32)
        if nd.primCnt:
            assert nd.primCnt <= threshold
            slots = range(nd.offset, nd.offset + nd.primCnt)
            p = np.concatenate([pos[[tris[order[k]].i[c] for c in range(3)]] for k in slots])
            assert np.array_equal(p.min(0), lo) and np.array_equal(p.max(0), hi), "leaf box is not the box of its triangles"
            seen.extend(order[k] for k in slots)
            cost += area(lo, hi) * nd.primCnt
        else:
            a, b = nodes[i + 1], nodes[nd.offset]
            assert nd.offset > i + 1 and nd.axis in (0, 1, 2)
            assert np.array_equal(np.minimum(a.bbmin[:], b.bbmin[:]).astype(np.float32), lo)
            assert np.array_equal(np.maximum(a.bbmax[:], b.bbmax[:]).astype(np.float32), hi)
            cost += area(lo, hi)
            stack += [i + 1, nd.offset]
    assert visited == cnt and sorted(seen) == list(range(len(tris)))
    return cost


def test_capi_median_build_is_the_host_builder(rtlib, oracle):
    verts, tris = random_mesh(700, seed=21)
    for tight in (0, 1):
        nodes, cnt, order = capi_build(rtlib, verts, tris, 0, 10, tight)
        ref = build_both(rtlib, oracle, verts, tris, 10, tight)["oracle"]
        assert bytes(nodes)[: 32 * cnt] == ref[0] and order == ref[1] and cnt == ref[2]


def test_surface_area_tree_is_valid_and_cheaper(rtlib):
    verts, tris = clustered_mesh(3000, seed=4)
    median = walk(*capi_build(rtlib, verts, tris, 0, 8, 1), verts, tris, 8)
    sah = walk(*capi_build(rtlib, verts, tris, 1, 8, 1), verts, tris, 8)
    print("surface-area cost: median cut", median, "binned SAH", sah)
    assert sah < 0.8 * median


def big_mesh(n_tris, seed):
    """Like clustered_mesh, filled in through numpy: large enough for the builder to use several threads."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-40, 40, (6, 3))
    spread = np.array([0.3, 0.5, 1.0, 2.0, 4.0, 30.0])
    k = rng.integers(0, 6, n_tris)
    c = centres[k] + rng.normal(0, 1, (n_tris, 3)) * spread[k, None]
    assert C.sizeof(N.RtVertex) == 56 and N.RtVertex.position.offset % 4 == 0
    at = N.RtVertex.position.offset // 4
    v = np.zeros((3 * n_tris, 14), dtype=np.float32)
    v[:, at : at + 3] = (c[:, None, :] + rng.uniform(-0.2, 0.2, (n_tris, 3, 3))).reshape(-1, 3)
    t = np.arange(3 * n_tris, dtype=np.int32).reshape(n_tris, 3).copy()
    verts = (N.RtVertex * (3 * n_tris)).from_buffer(v)
    tris = (N.RtTriangle * n_tris).from_buffer(t)
    return verts, tris, (v, t)


@pytest.mark.parametrize("rule", [0, 1])
def test_tree_does_not_depend_on_the_number_of_builder_threads(rtlib, oracle, rule, monkeypatch):
    """Ranges of 32768 primitives and more hand one half to another thread; nodes and leaf order must not change with it."""
    n = 150_000
    verts, tris, keep = big_mesh(n, seed=12)
    built = {}
    for threads in ("1", "2", "8"):
        monkeypatch.setenv("RT_BVH_THREADS", threads)
        nodes, cnt, order = capi_build(rtlib, verts, tris, rule, 8, 1)
        built[threads] = (bytes(nodes)[: 32 * cnt], order)
    assert built["1"] == built["2"] == built["8"]
    assert sorted(built["8"][1]) == list(range(n))
    if rule == 0:
        ref = build_both(rtlib, oracle, verts, tris, 8, 1)["oracle"]
        assert built["8"] == (ref[0], ref[1])


def test_bvh_build_rejects_bad_arguments(rtlib):
    verts, tris = random_mesh(4, seed=1)
    nodes = (N.RtLinearNode * 8)()
    cnt = C.c_int()
    order = (C.c_int * 4)()
    args = lambda **kw: [kw.get("v", C.addressof(verts)), 12, C.addressof(tris), kw.get("n", 4), kw.get("rule", 1), kw.get("thr", 1), 1,
                         C.addressof(nodes), kw.get("cap", 8), C.byref(cnt), C.addressof(order)]
    assert rtlib.rt_mesh_bvh_build(*args(v=None)) == -1
    assert rtlib.rt_mesh_bvh_build(*args(rule=7)) == -1
    assert rtlib.rt_mesh_bvh_build(*args(thr=0)) == -1
    assert rtlib.rt_mesh_bvh_build(*args(cap=3)) == -3 and b"node array too small" in rtlib.rt_last_error(None)
    tris[2].i[1] = 99
    assert rtlib.rt_mesh_bvh_build(*args()) == -1 and b"vertex index" in rtlib.rt_last_error(None)


@pytest.mark.parametrize("name,size", [("soup", 3000), ("mixed", 0), ("heightfield", 5000)])
def test_surface_area_trees_give_the_same_image(oracle, name, size):
    """Same hit keys and the same radiance with mesh BVHs cut by the surface-area heuristic (no exact ties in these
    scenes: nothing is coplanar along a ray), with fewer or as many box tests."""
    w, h = 64, 40
    a = rt.HostScene(name, w, h, size, 2, 2, tight_bvh=True)
    b = rt.HostScene(name, w, h, size, 2, 2, tight_bvh=True, sah_bvh=True)
    ra, ka, ca = oracle.render(a.desc, a.opts, a.camera, w, h, threads=4)
    rb, kb, cb = oracle.render(b.desc, b.opts, b.camera, w, h, threads=4)
    assert np.array_equal(ka, kb)
    assert np.array_equal(ra, rb, equal_nan=True)
    print(name, "box tests", ca["node"], "->", cb["node"], "triangle tests", ca["tri"], "->", cb["tri"])


@pytest.mark.parametrize("name,size", [("cubes", 6), ("soup", 3000), ("mixed", 0)])
def test_image_does_not_depend_on_the_bvh(oracle, name, size):
    """Tight vs reference bounds, and BVH vs linear scans (useSceneBvh / useMeshBvhs off), give the same hits."""
    w, h = 64, 40
    frames = []
    for tight in (False, True):
        s = rt.HostScene(name, w, h, size, 2, 2, tight_bvh=tight)
        frames.append(oracle.render(s.desc, s.opts, s.camera, w, h, threads=4)[:2])
    s = rt.HostScene(name, w, h, size, 2, 2, tight_bvh=True)
    s.opts.contents.useSceneBvh = 0
    s.opts.contents.useMeshBvhs = 0
    frames.append(oracle.render(s.desc, s.opts, s.camera, w, h, threads=4)[:2])
    for rgba, keys in frames[1:]:
        assert np.array_equal(keys, frames[0][1])
        assert np.array_equal(rgba, frames[0][0], equal_nan=True)


def test_any_hit_shadow_rays_give_the_same_image(oracle):
    s = rt.HostScene("mixed", 80, 60, 0, 5, 3)
    a = oracle.render(s.desc, s.opts, s.camera, 80, 60, threads=4, shadow_any_hit=False)
    b = oracle.render(s.desc, s.opts, s.camera, 80, 60, threads=4, shadow_any_hit=True)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert b[2]["node"] < a[2]["node"]  # fewer tests, same booleans


def test_matrix_inverse_host_equals_oracle(rtlib, oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        m = rng.uniform(-3, 3, 16).astype(np.float32)
        a = (C.c_float * 16)(*m)
        o1 = (C.c_float * 16)()
        o2 = (C.c_float * 16)()
        rtlib.rth_mat_inverse(a, o1)
        oracle.load().oracle_mat_inverse(a, o2)
        assert bytes(o1) == bytes(o2)
    ident = (C.c_float * 16)(*np.eye(4, dtype=np.float32).ravel())
    sing = (C.c_float * 16)(*np.zeros(16, dtype=np.float32))
    out = (C.c_float * 16)()
    rtlib.rth_mat_inverse(sing, out)  # singular -> identity (Mat.cuh:193-196)
    assert bytes(out) == bytes(ident)


def test_o3_build_of_the_oracle_renders_the_same_scene(oracle):
    """The -O3 / FMA build that the bench's CPU baseline also times is the same algorithm: its frames agree with the
    parity build's within the parity tolerance on nearly every pixel (contraction changes roundings, so not on all)."""
    if not oracle.o3_supported():
        pytest.skip("CPU without AVX2 / FMA / BMI2")
    from parity import compare

    scene = rt.HostScene("mixed", 160, 120, 0, 3, 3)
    a = oracle.render(scene.desc, scene.opts, scene.camera, 160, 120, threads=4)
    b = oracle.render(scene.desc, scene.opts, scene.camera, 160, 120, threads=4, o3=True)
    r = compare(b[0], b[1], a[0], a[1])
    assert r["key_agreement"] >= 0.999 and r["radiance_ok_fraction"] >= 0.999, r
    assert abs(b[2]["node"] - a[2]["node"]) <= 0.001 * a[2]["node"]
