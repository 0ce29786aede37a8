"""Mesh BVHs cut by the binned surface-area heuristic ON THE DEVICE (rt_set_device_bvh_rule) against the host builder with
the same rule (csrc/host/BVH.h sahPartition through rt_mesh_bvh_build): the blobs must be identical byte for byte -
same nodes, same boxes, same leaf order - and frames must match the oracle."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity
from test_gpu_device_bvh import mesh_sections

pytestmark = pytest.mark.gpu


def blob_of(ctx, dscene):
    ptr, n = ctx.scene_blob(dscene)
    return ctx.download(ptr, (n,), np.uint8)


@pytest.mark.parametrize("name,size,threshold", [("soup", 50000, 8), ("soup", 20011, 4), ("soup", 7, 8), ("soup", 11, 8), ("soup", 300, 1),
                                                 ("heightfield", 30000, 8), ("heightfield", 200000, 10), ("cubes", 6, 8), ("soup", 1000000, 8)])
def test_device_cut_tree_is_the_host_builders_tree(gpu_ctx, name, size, threshold):
    w, h = 64, 48
    host = rt.HostScene(name, w, h, size, 1, 0, tight_bvh=True, sah_bvh=True, leaf_threshold=threshold)
    dev = rt.HostScene(name, w, h, size, 1, 0, tight_bvh=True, device_bvh=True)
    gpu_ctx.set_device_bvh(threshold, True)
    gpu_ctx.set_device_bvh_rule(sah=True)
    a = b = None
    try:
        a = gpu_ctx.upload(host)
        b = gpu_ctx.upload(dev)
        blob_a, blob_b = blob_of(gpu_ctx, a), blob_of(gpu_ctx, b)
        assert len(blob_a) == len(blob_b)
        if name == "cubes":
            # indexed meshes: the generator forms per-face tangents in triangle order (cudaTracer/CudaMesh.h:106-130), which the
            # host-built scene has reordered by then - the vertex arrays differ in the sign of zeros; trees and records must not
            assert np.array_equal(blob_a[:256], blob_b[:256])
            for mesh in range(size * size):
                nodes_a, tris_a, _ = mesh_sections(blob_a, mesh)
                nodes_b, tris_b, _ = mesh_sections(blob_b, mesh)
                assert np.array_equal(nodes_a, nodes_b) and np.array_equal(tris_a, tris_b), mesh
        else:
            assert np.array_equal(blob_a, blob_b)
    finally:
        gpu_ctx.set_device_bvh_rule(sah=False)
        gpu_ctx.set_device_bvh(10, True)
        for s in (a, b):
            if s is not None:
                gpu_ctx.free_scene(s)


@pytest.mark.parametrize("name,size,depth", [("soup", 100000, 0), ("mixed", 0, 3), ("cubes", 8, 2)])
def test_frames_match_the_oracle(gpu_ctx, oracle, synthetic_assets, name, size, depth):
    """The handed-over trees (median cut, reference bounds) are thrown away at upload and cut again by surface area on
    the device (rt_set_upload_rebuild + rt_set_device_bvh_rule): every hit key as the oracle has it on the caller's trees."""
    w, h = 160, 96
    scene = rt.HostScene(name, w, h, size, 2, depth, asset_dir=synthetic_assets if name == "mixed" else None, tight_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    gpu_ctx.set_device_bvh(8, True)
    gpu_ctx.set_device_bvh_rule(sah=True)
    gpu_ctx.set_upload_rebuild(True)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    finally:
        gpu_ctx.set_upload_rebuild(False)
        gpu_ctx.set_device_bvh_rule(sah=False)
        gpu_ctx.set_device_bvh(10, True)
    assert_parity(rgba, keys, ref_rgba, ref_keys, radiance_fraction=1.0 if name == "soup" else 0.9998)
    assert (cnt["primary"], cnt["shadow"], cnt["secondary"]) == (ref_cnt["primary"], ref_cnt["shadow"], ref_cnt["secondary"])


def test_reference_bounds_are_refused(gpu_ctx):
    dev = rt.HostScene("soup", 32, 32, 1000, 1, 0, tight_bvh=True, device_bvh=True)
    gpu_ctx.set_device_bvh(8, False)
    gpu_ctx.set_device_bvh_rule(sah=True)
    try:
        with pytest.raises(rt.RtError):
            gpu_ctx.upload(dev)
    finally:
        gpu_ctx.set_device_bvh_rule(sah=False)
        gpu_ctx.set_device_bvh(10, True)


def test_clumps_of_coinciding_centroids(gpu_ctx, rtlib):
    """Triangles whose boxes - hence centroids - coincide: the host builder with the surface-area rule stops at a range
    whose centroids all coincide along the widest axis and makes it one leaf, whatever its size (csrc/host/BVH.h, after
    cudaTracer/BVH.h:126-128).  The device makes the same tree: the two scenes' blobs are identical."""
    import hand_scenes as hs
    from test_gpu_fuzz import bvh_of

    rng = np.random.default_rng(5)
    tris = []
    for c in ((-1.5, 0.0, 6.0), (1.5, 0.5, 7.0), (0.0, -1.0, 9.0)):
        c = np.float32(c)
        for k in range(25):  # 25 triangles per clump, all with the box [c - 1, c + 1]
            a, b = rng.uniform(-1, 1, 2)
            p = np.float32([[-1, a, -1], [1, -1, b], [b, 1, 1]]) + c
            tris.append(hs.tri(*[tuple(float(x) for x in v) for v in p]))
    for k in range(40):
        c = rng.uniform(-3, 3, 3) + np.float32([0, 0, 8])
        p = np.float32(c + rng.uniform(-0.7, 0.7, (3, 3)))
        tris.append(hs.tri(*[tuple(float(x) for x in v) for v in p]))
    ordered, spec = bvh_of(rtlib, tris, 1, 8, 1)  # RT_BVH_BINNED_SAH, leaves of 8, tight bounds
    assert max(n[3] for n in spec) > 8  # a clump stayed one leaf
    lights = [((0, 4, 0), (1, 1, 1), 8.0)]
    host_scene = hs.Scene(32, 32, [hs.Mesh(ordered, nodes=spec)], lights=lights)
    # the caller's order must not matter to the tree's shape; leaf slots follow the stable partition of THIS order
    dev_scene = hs.Scene(32, 32, [hs.Mesh(tris, device_bvh=True)], lights=lights)
    gpu_ctx.set_device_bvh(8, True)
    gpu_ctx.set_device_bvh_rule(sah=True)
    a = b = None
    try:
        import ctypes as C
        a = gpu_ctx.upload(C.pointer(host_scene.desc))
        b = gpu_ctx.upload(C.pointer(dev_scene.desc))
        blob_a, blob_b = blob_of(gpu_ctx, a), blob_of(gpu_ctx, b)
        nodes_a, tris_a, _ = mesh_sections(blob_a)
        nodes_b, tris_b, _ = mesh_sections(blob_b)
        assert np.array_equal(nodes_a, nodes_b)
        # triangle records: position and edges (the vertex numbers differ: hand_scenes numbers vertices by list position)
        assert np.array_equal(tris_a[:, :9], tris_b[:, :9])
    finally:
        gpu_ctx.set_device_bvh_rule(sah=False)
        gpu_ctx.set_device_bvh(10, True)
        for s in (a, b):
            if s is not None:
                gpu_ctx.free_scene(s)


def test_small_meshes_and_extreme_thresholds(gpu_ctx):
    """One to a few hundred triangles, leaves of 1, 2 and 63: a single leaf, a root with two leaves, ranges that fit two leaves
    (cut where the two boxes' surface is smallest) - blob for blob the host builder's scene."""
    try:
        for name in ("soup", "heightfield"):
            for size in (1, 2, 3, 9, 17, 257):
                for threshold in (1, 2, 63):
                    host = rt.HostScene(name, 32, 32, size, 1, 0, tight_bvh=True, sah_bvh=True, leaf_threshold=threshold)
                    dev = rt.HostScene(name, 32, 32, size, 1, 0, tight_bvh=True, device_bvh=True)
                    gpu_ctx.set_device_bvh(threshold, True)
                    gpu_ctx.set_device_bvh_rule(sah=True)
                    a, b = gpu_ctx.upload(host), gpu_ctx.upload(dev)
                    try:
                        assert np.array_equal(blob_of(gpu_ctx, a), blob_of(gpu_ctx, b)), (name, size, threshold)
                    finally:
                        gpu_ctx.free_scene(a)
                        gpu_ctx.free_scene(b)
    finally:
        gpu_ctx.set_device_bvh_rule(sah=False)
        gpu_ctx.set_device_bvh(10, True)
