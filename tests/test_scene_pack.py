"""rt_scene_pack on the host (no GPU): blob structure, the sibling-pair node layout
against the depth-first LinearNode input, and rejection of malformed scenes."""
import ctypes as C
import struct

import numpy as np
import pytest

import cudaraytracer_amd as rt
from cudaraytracer_amd import _native as N

RT_ERR_INVALID, RT_ERR_LIMIT = -1, -3
LEAF = 0x80000000


def header(blob):
    f = struct.unpack_from("<IIQ8i5QII", blob, 0)
    names = ["magic", "version", "totalBytes", "lightCnt", "shapeCnt", "meshCnt", "sceneNodeCnt", "sceneDepth", "maxMeshDepth",
             "hasSecondary", "maxLeafShapes", "offLights", "offShapes", "offSceneNodes", "offMeshes", "offRecords", "sceneRootRef",
             "recordCnt"]
    return dict(zip(names, f))


def pairs_to_dfs(blob, root_ref, prim_base, shapes_leaf):
    """Walks a sibling-pair tree from its root pseudo reference in the input's order (node, first child subtree, second
    child subtree) and returns [(bbmin, bbmax, is_leaf, first, count, axis)] so it can be compared with the depth-first input."""
    out = []
    assert (root_ref >> 29) == 3  # pseudo reference to a root record

    def rec(byte_off):
        f = np.frombuffer(blob[byte_off: byte_off + 24].tobytes(), dtype=np.float32)
        ref = struct.unpack_from("<I", blob[byte_off + 24: byte_off + 28].tobytes())[0]
        return f, ref

    def walk(byte_off):
        f, ref = rec(byte_off)
        if ref & LEAF:
            assert bool(ref & 0x40000000) == shapes_leaf
            out.append((f[:3].tolist(), f[3:].tolist(), True, (ref & 0xFFFFFF) - prim_base, (ref >> 24) & 0x3F, 0))
        else:
            out.append((f[:3].tolist(), f[3:].tolist(), False, 0, 0, (ref >> 29) & 3))
            pair = (ref & 0x1FFFFFFF) * 64  # absolute: 64-byte units from the blob start
            walk(pair)
            walk(pair + 32)

    walk((root_ref & 0x1FFFFFFF) * 64)
    return out


def dfs_input(nodes, cnt):
    out = []
    for i in range(cnt):
        n = nodes[i]
        leaf = n.primCnt > 0
        out.append((list(n.bbmin), list(n.bbmax), leaf, n.offset if leaf else 0, n.primCnt if leaf else 0, 0 if leaf else n.axis))
    return out


@pytest.mark.parametrize("name,size,tight", [("cubes", 5, False), ("mixed", 0, True), ("soup", 3000, True)])
def test_blob_holds_the_scene(rtlib, name, size, tight):
    s = rt.HostScene(name, 64, 48, size, 3, 2, tight_bvh=tight)
    d = s.desc.contents
    blob = rt.pack_scene(rtlib, s.desc)
    h = header(blob)
    assert h["magic"] == 0x52544246 and h["totalBytes"] == len(blob) and len(blob) % 256 == 0
    assert (h["lightCnt"], h["shapeCnt"], h["meshCnt"], h["sceneNodeCnt"]) == (d.lightCnt, d.shapeCnt, d.meshCnt, d.nodeCnt)
    assert h["version"] == 4 and h["recordCnt"] == sum(d.meshes[i].triangleCnt for i in range(d.meshCnt))
    for k in ("offLights", "offShapes", "offSceneNodes", "offMeshes", "offRecords"):
        assert h[k] % 256 == 0 and h[k] < len(blob)
    # lights verbatim
    assert blob[h["offLights"]: h["offLights"] + 28 * d.lightCnt].tobytes() == C.string_at(d.lights, 28 * d.lightCnt)
    # scene BVH: same boxes, leaves and split axes in the same depth-first order
    assert (h["sceneRootRef"] & 0x1FFFFFFF) * 64 == h["offSceneNodes"]
    assert pairs_to_dfs(blob, h["sceneRootRef"], 0, True) == dfs_input(d.nodes, d.nodeCnt)
    record_base = 0
    # every mesh BVH likewise, and triangle records = (v0, v1 - v0, v2 - v0, indices) in fp32
    for mi in range(d.meshCnt):
        m = d.meshes[mi]
        dm = struct.unpack_from("<16f6Q9i3i2I", blob, h["offMeshes"] + 192 * mi)
        off_nodes, off_tris, off_verts = dm[16], dm[17], dm[18]
        assert dm[31:34] == (m.triangleCnt, m.nodeCnt, m.vertexCnt)
        root_ref, rec_base = dm[34], dm[35]
        assert rec_base == record_base and off_tris == h["offRecords"] + 48 * rec_base
        record_base += m.triangleCnt
        assert (root_ref & 0x1FFFFFFF) * 64 == off_nodes
        assert pairs_to_dfs(blob, root_ref, rec_base, False) == dfs_input(m.nodes, m.nodeCnt)
        assert blob[off_verts: off_verts + 56 * m.vertexCnt].tobytes() == C.string_at(m.vertices, 56 * m.vertexCnt)
        rec = np.frombuffer(blob, dtype=np.float32, count=12 * m.triangleCnt, offset=off_tris).reshape(-1, 12)
        idx = np.frombuffer(C.string_at(m.triangles, 12 * m.triangleCnt), dtype=np.int32).reshape(-1, 3)
        pos = np.frombuffer(C.string_at(m.vertices, 56 * m.vertexCnt), dtype=np.float32).reshape(-1, 14)[:, 11:14]
        assert np.array_equal(rec[:, 0:3], pos[idx[:, 0]])
        assert np.array_equal(rec[:, 3:6], pos[idx[:, 1]] - pos[idx[:, 0]])
        assert np.array_equal(rec[:, 6:9], pos[idx[:, 2]] - pos[idx[:, 0]])
        assert np.array_equal(rec[:, 9:12].view(np.int32), idx)
    # packing is deterministic (the blob is what gets broadcast to the other GPUs)
    assert np.array_equal(blob, rt.pack_scene(rtlib, s.desc))


def expect_error(rtlib, desc, code, text):
    p, n = C.c_void_p(), C.c_size_t()
    rc = rtlib.rt_scene_pack(None, desc, C.byref(p), C.byref(n))
    assert rc == code, (rc, rtlib.rt_last_error(None))
    assert text in rtlib.rt_last_error(None).decode()


def test_malformed_scenes_are_rejected(rtlib):
    s = rt.HostScene("mixed", 32, 32, 0, 2, 2)
    d = s.desc.contents
    mesh = d.meshes[0]
    # triangle index out of range
    keep = mesh.triangles[0].i[1]
    mesh.triangles[0].i[1] = mesh.vertexCnt
    expect_error(rtlib, s.desc, RT_ERR_INVALID, "vertex index")
    mesh.triangles[0].i[1] = keep
    # leaf reaching past the primitive list
    leaf = next(i for i in range(mesh.nodeCnt) if mesh.nodes[i].primCnt > 0)
    keep = mesh.nodes[leaf].offset
    mesh.nodes[leaf].offset = mesh.triangleCnt
    expect_error(rtlib, s.desc, RT_ERR_INVALID, "")
    mesh.nodes[leaf].offset = keep
    # interior node whose second child points outside the array
    inner = next((i for i in range(d.nodeCnt) if d.nodes[i].primCnt == 0), None)
    if inner is not None:
        keep = d.nodes[inner].offset
        d.nodes[inner].offset = d.nodeCnt + 5
        expect_error(rtlib, s.desc, RT_ERR_INVALID, "")
        d.nodes[inner].offset = keep
    # nodes the root does not reach: an interior node of the mesh turned into a leaf strands its subtree
    inner = next((i for i in range(mesh.nodeCnt) if mesh.nodes[i].primCnt == 0), None)
    if inner is not None:
        keep = (mesh.nodes[inner].offset, mesh.nodes[inner].primCnt)
        mesh.nodes[inner].offset, mesh.nodes[inner].primCnt = 0, 1
        expect_error(rtlib, s.desc, RT_ERR_INVALID, "does not reach")
        mesh.nodes[inner].offset, mesh.nodes[inner].primCnt = keep
    # a child box that sticks out of its parent's
    if inner is not None:
        child = inner + 1
        keep = mesh.nodes[child].bbmax[0]
        mesh.nodes[child].bbmax[0] = mesh.nodes[inner].bbmax[0] + 1.0
        expect_error(rtlib, s.desc, RT_ERR_INVALID, "does not contain")
        mesh.nodes[child].bbmax[0] = keep
    # shape naming a mesh that is not there
    ms = next(i for i in range(d.shapeCnt) if d.shapes[i].kind == 1)
    keep = d.shapes[ms].meshIdx
    d.shapes[ms].meshIdx = d.meshCnt
    expect_error(rtlib, s.desc, RT_ERR_INVALID, "mesh index")
    d.shapes[ms].meshIdx = keep
    # empty scene
    keep = d.shapeCnt
    d.shapeCnt = 0
    expect_error(rtlib, s.desc, RT_ERR_INVALID, "without shapes")
    d.shapeCnt = keep
    # and the untouched description still packs
    assert len(rt.pack_scene(rtlib, s.desc)) > 0


def test_leaf_larger_than_the_reference_field_allows_is_a_limit_error(rtlib):
    """A one-leaf BVH over 200 triangles (what the reference builds for coincident centroids) exceeds the 6-bit
    leaf count of the packed node reference."""
    s = rt.HostScene("soup", 32, 32, 200, 1, 0)
    d = s.desc.contents
    m = d.meshes[0]
    nodes = (N.RtLinearNode * 1)()
    nodes[0].bbmin[:] = [-100.0] * 3
    nodes[0].bbmax[:] = [100.0] * 3
    nodes[0].offset, nodes[0].primCnt = 0, 200
    keep = (m.nodes, m.nodeCnt)
    m.nodes, m.nodeCnt = C.cast(nodes, C.POINTER(N.RtLinearNode)), 1
    expect_error(rtlib, s.desc, RT_ERR_LIMIT, "")
    m.nodes, m.nodeCnt = keep


def test_chain_deeper_than_the_traversal_stack_is_a_limit_error(rtlib):
    s = rt.HostScene("soup", 32, 32, 200, 1, 0)
    d = s.desc.contents
    m = d.meshes[0]
    depth = N.RT_STACK_ENTRIES + 4
    cnt = 2 * depth + 1
    nodes = (N.RtLinearNode * cnt)()
    # left-deep chain: interior i has a leaf as second child at the end of the array
    for i in range(depth):
        nodes[i].bbmin[:] = [-100.0] * 3
        nodes[i].bbmax[:] = [100.0] * 3
        nodes[i].primCnt, nodes[i].axis = 0, 0
        nodes[i].offset = cnt - 1 - i  # second child
    for j in range(depth, cnt):
        nodes[j].bbmin[:] = [-100.0] * 3
        nodes[j].bbmax[:] = [100.0] * 3
        nodes[j].primCnt, nodes[j].offset = 1, (j - depth) % 200
    keep = (m.nodes, m.nodeCnt)
    m.nodes, m.nodeCnt = C.cast(nodes, C.POINTER(N.RtLinearNode)), cnt
    expect_error(rtlib, s.desc, RT_ERR_LIMIT, "too deep")
    m.nodes, m.nodeCnt = keep


def test_host_only_packing_rejects_a_mesh_without_nodes(rtlib):
    """Meshes without BVH nodes are for rt_scene_upload, which builds their BVH on the device."""
    s = rt.HostScene("soup", 32, 32, 500, 1, 0, tight_bvh=True, device_bvh=True)
    m = s.desc.contents.meshes[0]
    assert m.nodeCnt == 0 and not m.nodes
    expect_error(rtlib, s.desc, RT_ERR_INVALID, "without BVH nodes")


def test_tile_row_and_column_by_multiplication():
    """The kernels split a tile index into row and column with P.tilesXMagic = floor((2^32 - 1) / tilesX): the high word of
    tile * magic is the row or one less, one comparison settles it (rt_kernels.hip slotPixel).  The same arithmetic here, for
    every tile count per row up to 4096 at the values where an estimate could be off, and for all tiles of a few frames."""
    import numpy as np

    def split(tile, d):
        magic = np.uint64(0xFFFFFFFF // d)
        row = (tile.astype(np.uint64) * magic) >> np.uint64(32)
        col = tile.astype(np.int64) - row.astype(np.int64) * d
        fix = col >= d
        return row.astype(np.int64) + fix, col - fix * d

    for d in range(1, 4097):
        ks = np.array([0, 1, 2, 1000, 4095, 4096, (1 << 20) // d, (1 << 24) // d, (1 << 31) // d - 1], dtype=np.int64)
        tiles = np.unique(np.clip(np.concatenate([ks * d - 1, ks * d, ks * d + 1, ks * d + d - 1]), 0, (1 << 31) - 1)).astype(np.uint32)
        row, col = split(tiles, d)
        assert np.array_equal(row, tiles.astype(np.int64) // d) and np.array_equal(col, tiles.astype(np.int64) % d), d
    for d in (1, 3, 80, 240, 480, 960, 1000):
        tiles = np.arange(0, 1 << 21, dtype=np.uint32)
        row, col = split(tiles, d)
        assert np.array_equal(row, tiles.astype(np.int64) // d) and np.array_equal(col, tiles.astype(np.int64) % d), d
