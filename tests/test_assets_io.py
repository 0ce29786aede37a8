"""Asset ingest and frame dump on the host side: PNG codec (Png.h, standing in for
the reference's lodepng calls at cudaTracer/RayTracer.cpp:99-117 and PicDumper.cpp:19-29),
OBJ reader (ObjFileReader.cpp behaviour) and the float -> 8-bit rule (PicDumper.cpp:31-45)."""
import ctypes as C
import os

import numpy as np
import pytest

from cudaraytracer_amd import _native as N

PIL = pytest.importorskip("PIL.Image")
REF_ASSETS = "/root/reference/assets"


def decode(rtlib, path):
    w, h = C.c_int(), C.c_int()
    assert rtlib.rth_read_png(str(path).encode(), None, 0, C.byref(w), C.byref(h)) == 0, rtlib.rth_last_error()
    buf = (C.c_ubyte * (w.value * h.value * 4))()
    assert rtlib.rth_read_png(str(path).encode(), buf, len(buf), C.byref(w), C.byref(h)) == 0
    return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, w.value, 4).copy()


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1", "I;16"])
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 33)])
def test_png_decode_matches_pillow(rtlib, tmp_path, mode, size):
    rng = np.random.default_rng(hash((mode, size)) & 0xFFFF)
    w, h = size
    if mode == "I;16":
        arr = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        img = PIL.fromarray(arr)
        want = np.stack([(arr >> 8).astype(np.uint8)] * 3 + [np.full((h, w), 255, np.uint8)], axis=-1)  # 16 -> 8 bit: high byte
    else:
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        img = PIL.fromarray(rgba, mode="RGBA").convert(mode)
        want = np.asarray(img.convert("RGBA"))
    path = tmp_path / f"t_{mode.replace(';', '')}.png"
    img.save(path)
    got = decode(rtlib, path)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_png_decode_interlaced_and_transparency(rtlib, tmp_path):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    img = PIL.fromarray(rgb, mode="RGB")
    p = tmp_path / "trns.png"
    key = tuple(int(v) for v in rgb[4, 5])
    img.save(p, transparency=key)
    got = decode(rtlib, p)
    want = np.asarray(PIL.open(p).convert("RGBA"))
    assert np.array_equal(got, want)
    assert got[4, 5, 3] == 0


def _chunk(kind, body):
    import struct
    import zlib

    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (9, 9), (33, 18)])
def test_png_decode_adam7_interlaced(rtlib, tmp_path, size):
    """Pillow does not write interlaced files, so the seven passes are laid out by hand (filter 0 rows)."""
    import struct
    import zlib

    w, h = size
    rng = np.random.default_rng(w * 100 + h)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    raw = b""
    for p in range(7):
        sub = img[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\0" + row.tobytes()
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 1)) + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b"")
    path = tmp_path / "adam7.png"
    path.write_bytes(data)
    assert np.array_equal(np.asarray(PIL.open(path).convert("RGBA")), img)  # the hand-made file is a valid PNG
    assert np.array_equal(decode(rtlib, path), img)


def test_png_decode_rejects_garbage(rtlib, tmp_path):
    p = tmp_path / "bad.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    w, h = C.c_int(), C.c_int()
    assert rtlib.rth_read_png(str(p).encode(), None, 0, C.byref(w), C.byref(h)) != 0
    assert rtlib.rth_read_png(str(tmp_path / "missing.png").encode(), None, 0, C.byref(w), C.byref(h)) != 0


def test_frame_dump_bytes_rule_and_png_roundtrip(rtlib, tmp_path):
    """PicDumper.cpp:31-45: byte = clamp((int)(v * 256), 0, 255), alpha included."""
    vals = np.array([-1.0, -1e-9, 0.0, 0.0039, 0.00391, 0.5, 0.999, 0.99609375, 1.0, 7.5, 1e6, 1 / 256, 255 / 256, 0.25, 0.75, 0.1],
                    dtype=np.float32).reshape(2, 2, 4)
    out = (C.c_ubyte * vals.size)()
    rtlib.rth_floats_to_bytes(vals.ctypes.data, vals.size, out)
    want = np.clip((vals.astype(np.float64) * 256.0).astype(np.int64), 0, 255).astype(np.uint8)
    assert np.array_equal(np.frombuffer(out, dtype=np.uint8).reshape(2, 2, 4), want)
    p = tmp_path / "dump.png"
    assert rtlib.rth_write_png(str(p).encode(), vals.ctypes.data, 2, 2) == 0
    assert np.array_equal(np.asarray(PIL.open(p).convert("RGBA")), want)
    assert np.array_equal(decode(rtlib, p), want)


def test_png_encode_large_random_frame_roundtrip(rtlib, tmp_path):
    rng = np.random.default_rng(11)
    img = rng.random((120, 200, 4), dtype=np.float32)
    p = tmp_path / "big.png"
    assert rtlib.rth_write_png(str(p).encode(), img.ctypes.data, 200, 120) == 0
    want = np.clip((img * np.float32(256.0)).astype(np.int32), 0, 255).astype(np.uint8)
    assert np.array_equal(np.asarray(PIL.open(p).convert("RGBA")), want)


OBJ = """# comment
mtllib quad.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
vn 0 1 0
g first
usemtl mat
f 1/1/1 2/2/1 3/3/1 4/4/1
f 1/1/2 2/2/2 5/3/2
"""
MTL = """newmtl mat
Kd 1 1 1
map_Kd d.png
map_bump n.png
map_Ks s.png
"""


def obj_read(rtlib, folder, name, start_at_zero=0):
    counts = (C.c_int * 4)()
    names = C.create_string_buffer(4096)
    rc = rtlib.rth_obj_read(folder.encode(), name.encode(), start_at_zero, counts, None, 0, None, 0, names, len(names))
    assert rc == 0, rtlib.rth_last_error()
    verts = (N.RtVertex * max(counts[1], 1))()
    inds = (C.c_int * max(counts[2], 1))()
    assert rtlib.rth_obj_read(folder.encode(), name.encode(), start_at_zero, counts, verts, counts[1], inds, counts[2], names, len(names)) == 0
    return list(counts), verts, list(inds)[: counts[2]], names.value.decode().split("\n")


def test_obj_reader_quads_groups_and_materials(rtlib, tmp_path):
    folder = str(tmp_path) + "/"
    (tmp_path / "quad.obj").write_text(OBJ)
    (tmp_path / "quad.mtl").write_text(MTL)
    counts, verts, inds, names = obj_read(rtlib, folder, "quad.obj")
    assert counts[0] == 1
    assert counts[1] == 9 and counts[2] == 9  # quad -> 2 triangles (0,1,2)(0,2,3) + 1 triangle, un-indexed
    assert inds == list(range(9))
    pos = [tuple(verts[i].position) for i in range(9)]
    assert pos[:6] == [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 0), (1, 1, 0), (0, 1, 0)]
    assert pos[8] == (0.5, 0.5, 1.0)
    assert tuple(verts[0].normal) == (0, 0, 1) and tuple(verts[6].normal) == (0, 1, 0)
    assert tuple(verts[2].texCoord) == (1, 1)
    assert names == [folder + "d.png", folder + "n.png", folder + "s.png"]


def test_obj_reader_bad_index_is_an_error(rtlib, tmp_path):
    (tmp_path / "bad.obj").write_text("v 0 0 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n")
    counts = (C.c_int * 4)()
    assert rtlib.rth_obj_read((str(tmp_path) + "/").encode(), b"bad.obj", 0, counts, None, 0, None, 0, None, 0) == -1


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference assets not present on this machine")
def test_reference_barrel_asset_is_ingested(rtlib):
    folder = REF_ASSETS + "/models/metalBarrel/"
    counts, verts, inds, names = obj_read(rtlib, folder, "metal_barrel.obj")
    assert counts[0] >= 1 and counts[1] == counts[2] and counts[1] % 3 == 0 and counts[1] > 1000
    assert [os.path.basename(n) for n in names] == ["diffus_stainless_steel.png", "normal_hard_bumps.png", "specular_stainless_steel.png"]
    for n in names:
        px = decode(rtlib, n)
        want = np.asarray(PIL.open(n).convert("RGBA"))
        assert np.array_equal(px, want), n
