"""Mesh BVHs built on the GPU (rt_set_device_bvh / meshes uploaded without nodes) against the host builder
(csrc/host/BVH.h, bit-identical to cudaTracer/BVH.h) and against the oracle's images."""
import struct

import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity

pytestmark = pytest.mark.gpu
LEAF = 0x80000000


def blob_of(ctx, dscene):
    ptr, n = ctx.scene_blob(dscene)
    return ctx.download(ptr, (n,), np.uint8)


def mesh_sections(blob, mesh=0):
    off_meshes, off_records = struct.unpack_from("<QQ", blob, 72)
    dm = struct.unpack_from("<16f6Q9i3i2I", blob, off_meshes + 192 * mesh)
    off_nodes, off_tris = dm[16], dm[17]
    tri_cnt, node_cnt = dm[31], dm[32]
    interior = (node_cnt - 1) // 2
    recs = 2 * (interior + 1)
    nodes = np.frombuffer(blob, dtype=np.uint32, count=8 * recs, offset=off_nodes).reshape(recs, 8)
    tris = np.frombuffer(blob, dtype=np.uint32, count=12 * tri_cnt, offset=off_tris).reshape(tri_cnt, 12)
    return nodes, tris, dm[35]


def leaf_of_record(nodes, tri_cnt, rec_base):
    owner = np.full(tri_cnt, -1, dtype=np.int64)
    for k in range(len(nodes)):
        if k == 1:
            continue  # padding next to the root
        ref = int(nodes[k, 6])
        if ref & LEAF:
            first, count = (ref & 0xFFFFFF) - rec_base, (ref >> 24) & 0x3F
            owner[first:first + count] = k
    assert (owner >= 0).all()
    return owner


def check_tree(nodes, tris, rec_base, tight):
    """Every leaf box is the union of its triangles' corners, every interior box the union of its children, and the
    leaves partition the records."""
    f = nodes.view(np.float32)
    rec = tris.view(np.float32)
    v0, e1, e2 = rec[:, 0:3], rec[:, 3:6], rec[:, 6:9]
    corners = np.stack([v0, v0 + e1, v0 + e2], axis=1)  # e = v - v0 was rounded once; adding it back is exact enough to compare boxes loosely
    lo, hi = corners.min(axis=1), corners.max(axis=1)
    if not tight:
        hi = np.maximum(hi, np.finfo(np.float32).tiny)
    seen = np.zeros(len(tris), dtype=bool)
    base_unit = None

    def walk(k):
        nonlocal base_unit
        ref = int(nodes[k, 6])
        if ref & LEAF:
            first, count = (ref & 0xFFFFFF) - rec_base, (ref >> 24) & 0x3F
            assert count >= 1 and not seen[first:first + count].any()
            seen[first:first + count] = True
            assert np.allclose(f[k, 0:3], lo[first:first + count].min(axis=0), rtol=1e-6, atol=1e-6)
            assert np.allclose(f[k, 3:6], hi[first:first + count].max(axis=0), rtol=1e-6, atol=1e-6)
        else:
            unit = ref & 0x1FFFFFFF
            if base_unit is None:
                base_unit = unit - 1  # the root's children are the mesh's second pair
            pair = 2 * (unit - base_unit)
            assert ((ref >> 29) & 3) < 3
            walk(pair)
            walk(pair + 1)
            assert np.array_equal(f[k, 0:3], np.minimum(f[pair, 0:3], f[pair + 1, 0:3]))
            assert np.array_equal(f[k, 3:6], np.maximum(f[pair, 3:6], f[pair + 1, 3:6]))

    import sys
    sys.setrecursionlimit(10000)
    walk(0)
    assert seen.all()


@pytest.mark.parametrize("name,size,tight,exact", [("soup", 50000, True, True), ("soup", 20011, False, True), ("heightfield", 30000, True, False),
                                                    ("soup", 7, True, True), ("soup", 11, True, True), ("soup", 1000000, True, True)])
def test_device_built_bvh_equals_host_built_bvh(gpu_ctx, name, size, tight, exact):
    """`exact`: random geometry has distinct centroids, so the device's sort and the host's nth_element put the same
    triangles on either side of every cut and the two trees are bit-identical.  A regular grid has many equal centroid
    coordinates; there the cuts may differ (as between two nth_element implementations) and the tree is checked for
    validity instead."""
    w, h = 64, 48
    host = rt.HostScene(name, w, h, size, 1, 0, tight_bvh=tight)
    dev = rt.HostScene(name, w, h, size, 1, 0, tight_bvh=tight, device_bvh=True)
    gpu_ctx.set_device_bvh(10, tight)
    a = gpu_ctx.upload(host)
    b = gpu_ctx.upload(dev)
    try:
        blob_a, blob_b = blob_of(gpu_ctx, a), blob_of(gpu_ctx, b)
        assert len(blob_a) == len(blob_b)
        assert np.array_equal(blob_a[:256], blob_b[:256])  # same header: counts, depths, offsets
        nodes_a, tris_a, base = mesh_sections(blob_a)
        nodes_b, tris_b, _ = mesh_sections(blob_b)
        # every node record: box, reference (incl. split axis) identical
        if size <= 200000:
            check_tree(nodes_b, tris_b, base, tight)
        if not exact:
            return
        keep = np.ones(len(nodes_a), dtype=bool)
        keep[1] = False
        assert np.array_equal(nodes_a[keep][:, :7], nodes_b[keep][:, :7])
        # every leaf holds the same triangles (their order inside the leaf is free)
        own_a, own_b = leaf_of_record(nodes_a, len(tris_a), base), leaf_of_record(nodes_b, len(tris_b), base)
        assert np.array_equal(own_a, own_b)
        key_a = np.lexsort((tris_a[:, 11], tris_a[:, 10], tris_a[:, 9], own_a))
        key_b = np.lexsort((tris_b[:, 11], tris_b[:, 10], tris_b[:, 9], own_b))
        assert np.array_equal(tris_a[key_a], tris_b[key_b])
    finally:
        gpu_ctx.free_scene(a)
        gpu_ctx.free_scene(b)


@pytest.mark.parametrize("name,size,depth,tight", [("soup", 100000, 0, True), ("mixed", 0, 3, False), ("cubes", 8, 2, True)])
def test_frames_with_device_built_bvhs_match_the_oracle(gpu_ctx, oracle, name, size, depth, tight):
    w, h = 160, 96
    host = rt.HostScene(name, w, h, size, 2, depth, tight_bvh=tight)
    dev = rt.HostScene(name, w, h, size, 2, depth, tight_bvh=tight, device_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(host.desc, host.opts, host.camera, w, h, threads=8, shadow_any_hit=True)
    gpu_ctx.set_device_bvh(10, tight)
    rgba, keys, cnt = gpu_ctx.render_frame(dev, want_keys=True, want_counters=True)
    # A ray through an edge shared by two triangles hits both at the same distance and the later one in leaf order
    # wins (cudaTracer/Mesh.cuh:203-208): with another order inside the leaves such a pixel may change its key, and
    # per-face tangents of shared vertices follow the triangle order too (cudaTracer/CudaMesh.h:106-130).  Both only
    # concern indexed meshes (the cubes); the allowance is 1 pixel in 5000.
    assert_parity(rgba, keys, ref_rgba, ref_keys, radiance_fraction=1.0 if name == "soup" else 0.9998)
    # same tree, so the same box tests; a shadow ray stops at the first accepted triangle of a leaf, and the order
    # inside a leaf is the one thing the two builders do not share, so the triangle-test totals differ by a few
    # (cube meshes have many equal centroid coordinates: their cuts, hence a few box tests, may differ too)
    if name == "soup":
        assert cnt["node"] == ref_cnt["node"]
        assert abs(cnt["tri"] - ref_cnt["tri"]) <= 0.01 * ref_cnt["tri"]
    assert (cnt["primary"], cnt["shadow"], cnt["secondary"]) == (ref_cnt["primary"], ref_cnt["shadow"], ref_cnt["secondary"])


def test_other_leaf_sizes(gpu_ctx, oracle):
    w, h = 128, 80
    host = rt.HostScene("soup", w, h, 40000, 1, 0, tight_bvh=True)
    dev = rt.HostScene("soup", w, h, 40000, 1, 0, tight_bvh=True, device_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(host.desc, host.opts, host.camera, w, h, threads=8, shadow_any_hit=True)
    try:
        for threshold in (1, 4, 63):
            gpu_ctx.set_device_bvh(threshold, True)
            rgba, keys, cnt = gpu_ctx.render_frame(dev, want_keys=True, want_counters=True)
            assert_parity(rgba, keys, ref_rgba, ref_keys)  # hit keys do not depend on the tree
            assert (cnt["primary"], cnt["shadow"]) == (ref_cnt["primary"], ref_cnt["shadow"])
        with pytest.raises(rt.RtError):
            gpu_ctx.set_device_bvh(64, True)
    finally:
        gpu_ctx.set_device_bvh(10, True)


def test_ranges_of_coinciding_centroids_build_on_the_device(gpu_ctx, oracle, rtlib):
    """Triangles whose boxes - hence box centroids - coincide: the reference's builder stops at such a range and makes
    it one leaf of any size (cudaTracer/BVH.h:126-128); the device build cuts it at the median position.  Same hits:
    the frame of the device-built tree equals the oracle's on the host-built one."""
    import hand_scenes as hs
    from test_gpu_fuzz import bvh_of

    rng = np.random.default_rng(5)
    tris = []
    for c in ((-1.5, 0.0, 6.0), (1.5, 0.5, 7.0), (0.0, -1.0, 9.0)):
        c = np.float32(c)
        for k in range(25):  # 25 triangles per clump, all with the box [c - 1, c + 1]: each touches all six faces
            a, b = rng.uniform(-1, 1, 2)
            p = np.float32([[-1, a, -1], [1, -1, b], [b, 1, 1]]) + c
            if np.cross(p[1] - p[0], p[2] - p[0])[2] > 0:
                p = p[[0, 2, 1]]
            tris.append(hs.tri(*[tuple(float(x) for x in v) for v in p]))
    for k in range(40):  # and some ordinary ones around them
        c = rng.uniform(-3, 3, 3) + np.float32([0, 0, 8])
        p = np.float32(c + rng.uniform(-0.7, 0.7, (3, 3)))
        if np.cross(p[1] - p[0], p[2] - p[0])[2] > 0:
            p = p[[0, 2, 1]]
        tris.append(hs.tri(*[tuple(float(x) for x in v) for v in p]))
    lights = [((0, 4, 0), (1, 1, 1), 8.0), ((3, 0, 2), (1, 1, 1), 5.0)]
    w, h = 96, 64
    # the oracle walks the host builder's tree, in which a range that holds only triangles of one clump stays one leaf
    ordered, spec = bvh_of(rtlib, tris, 0, 10, 1)
    assert max(n[3] for n in spec) > 10
    host_scene = hs.Scene(w, h, [hs.Mesh(ordered, nodes=spec)], lights=lights)
    ref_rgba, ref_keys = host_scene.render_oracle(oracle)
    # hand_scenes numbers a triangle's vertices by its position in the list: give both scenes the leaf-ordered list
    dev_scene = hs.Scene(w, h, [hs.Mesh(ordered, device_bvh=True)], lights=lights)
    gpu_ctx.set_device_bvh(10, True)
    rgba, keys = dev_scene.render_gpu(gpu_ctx)
    assert (ref_keys[..., 0] == 0).mean() > 0.04
    assert np.array_equal(keys, ref_keys)
    assert np.array_equal(rgba, ref_rgba, equal_nan=True)
