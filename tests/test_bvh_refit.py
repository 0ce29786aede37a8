"""Bounds refit (rt_bvh_refit on the host; rt_set_upload_refit does the same on the device, tests/test_gpu_refit.py):
the reference's builder unions boxes from an "empty" box with max = FLT_MIN (cudaTracer/BoundingBox.cuh:25-27), so its
node arrays carry boxes that reach up to 0 on every axis; the refit keeps the tree and recomputes the boxes."""
import ctypes as C

import numpy as np
import pytest

import cudaraytracer_amd as rt
from cudaraytracer_amd import _native as N


def nodes_of(ptr, cnt):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(cnt * 32,)).view(
        np.dtype([("bbmin", "<f4", 3), ("bbmax", "<f4", 3), ("offset", "<i4"), ("primCnt", "<u2"), ("axis", "u1"), ("pad", "u1")]))


def refit_scene_in_place(lib, desc):
    """rt_bvh_refit on every tree of a scene description (the arrays belong to the host scene and are rewritten)."""
    d = desc.contents
    for i in range(d.meshCnt):
        m = d.meshes[i]
        assert lib.rt_bvh_refit(m.nodes, m.nodeCnt, m.vertices, m.vertexCnt, m.triangles, m.triangleCnt) == 0, lib.rt_last_error(None)
    assert lib.rt_scene_bvh_refit(d.nodes, d.nodeCnt, d.shapes, d.shapeCnt) == 0, lib.rt_last_error(None)


def test_refit_gives_exact_nested_boxes(rtlib):
    scene = rt.HostScene("soup", 64, 48, 5000, 1, 0, tight_bvh=False)
    m = scene.desc.contents.meshes[0]
    before = nodes_of(m.nodes, m.nodeCnt).copy()
    assert (before["bbmax"] >= 0).all() and (before["bbmax"] == np.float32(1.17549435e-38)).any()  # the FLT_MIN clamp at work
    assert rtlib.rt_bvh_refit(m.nodes, m.nodeCnt, m.vertices, m.vertexCnt, m.triangles, m.triangleCnt) == 0
    after = nodes_of(m.nodes, m.nodeCnt)
    # topology untouched
    for f in ("offset", "primCnt", "axis"):
        assert np.array_equal(before[f], after[f])
    # tighter or equal everywhere, strictly tighter somewhere
    assert (after["bbmin"] >= before["bbmin"]).all() and (after["bbmax"] <= before["bbmax"]).all()
    assert (after["bbmax"] < before["bbmax"]).any()
    pos = np.ctypeslib.as_array(C.cast(m.vertices, C.POINTER(C.c_float)), shape=(m.vertexCnt, 14))[:, 11:14]
    tri = np.ctypeslib.as_array(C.cast(m.triangles, C.POINTER(C.c_int32)), shape=(m.triangleCnt, 3))
    for i in range(m.nodeCnt):
        n = after[i]
        if n["primCnt"]:
            p = pos[tri[n["offset"]:n["offset"] + n["primCnt"]].ravel()]
            assert np.array_equal(n["bbmin"], p.min(0)) and np.array_equal(n["bbmax"], p.max(0))
        else:
            a, b = after[i + 1], after[n["offset"]]
            assert np.array_equal(n["bbmin"], np.minimum(a["bbmin"], b["bbmin"])) and np.array_equal(n["bbmax"], np.maximum(a["bbmax"], b["bbmax"]))
    # a second refit changes nothing, and neither does a refit of a tree built with true bounds
    again = after.copy()
    assert rtlib.rt_bvh_refit(m.nodes, m.nodeCnt, m.vertices, m.vertexCnt, m.triangles, m.triangleCnt) == 0
    assert np.array_equal(nodes_of(m.nodes, m.nodeCnt), again)
    tight = rt.HostScene("soup", 64, 48, 5000, 1, 0, tight_bvh=True)
    tm = tight.desc.contents.meshes[0]
    tb = nodes_of(tm.nodes, tm.nodeCnt).copy()
    assert rtlib.rt_bvh_refit(tm.nodes, tm.nodeCnt, tm.vertices, tm.vertexCnt, tm.triangles, tm.triangleCnt) == 0
    assert np.array_equal(nodes_of(tm.nodes, tm.nodeCnt), tb)


@pytest.mark.parametrize("name,size,lights,depth", [("soup", 20000, 2, 0), ("mixed", 0, 3, 3), ("cubes", 6, 2, 2)])
def test_refit_keeps_the_image_and_cuts_the_tests(rtlib, oracle, name, size, lights, depth):
    """Same hit keys and radiance on the reference's boxes and on the refitted ones (the oracle walks both): the tree,
    the visiting order and the tie rule are the same, only boxes that a ray cannot hit anything in are skipped."""
    w, h = 96, 64
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=False)
    rgba0, keys0, cnt0 = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    refit_scene_in_place(rtlib, scene.desc)
    rgba1, keys1, cnt1 = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    assert np.array_equal(keys0, keys1)
    assert np.array_equal(rgba0, rgba1, equal_nan=True)
    assert cnt1["node"] < cnt0["node"] and cnt1["tri"] <= cnt0["tri"]
    for k in ("primary", "shadow", "secondary", "shade_mesh", "shade_sphere"):
        assert cnt0[k] == cnt1[k]


def test_refit_rejects_bad_arguments(rtlib):
    scene = rt.HostScene("cubes", 32, 32, 2, 1, 0)
    m = scene.desc.contents.meshes[0]
    assert rtlib.rt_bvh_refit(None, 0, None, 0, None, 0) != 0
    assert rtlib.rt_scene_bvh_refit(None, 0, None, 0) != 0
    assert rtlib.rt_bvh_refit(m.nodes, m.nodeCnt, None, 0, m.triangles, m.triangleCnt) != 0  # triangles without vertices
    assert b"rt_bvh_refit" in rtlib.rt_last_error(None)
    assert rtlib.rt_bvh_refit(m.nodes, m.nodeCnt, m.vertices, m.vertexCnt, m.triangles, m.triangleCnt - 1) != 0  # leaves name triangles out of range
    assert N is not None
