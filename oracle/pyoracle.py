"""ctypes loader of oracle/liboracle.so - TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the cudaraytracer_amd package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `python -m cudaraytracer_amd.build`")
        lib = C.CDLL(LIB_PATH)
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int]
        lib.oracle_selftest.restype = C.c_int
        lib.oracle_selftest.argtypes = [C.c_char_p, C.c_int]
        lib.oracle_bvh_triangles.restype = C.c_int
        lib.oracle_bvh_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        lib.oracle_bvh_boxes.restype = C.c_int
        lib.oracle_bvh_boxes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        f3 = C.POINTER(C.c_float)
        lib.oracle_box_intersect.restype = C.c_int
        lib.oracle_box_intersect.argtypes = [f3, f3, f3, f3, C.c_float]
        lib.oracle_triangle_intersect.restype = C.c_int
        lib.oracle_triangle_intersect.argtypes = [f3, f3, f3, f3, f3, C.c_float, f3]
        lib.oracle_sphere_intersect.restype = C.c_int
        lib.oracle_sphere_intersect.argtypes = [f3, C.c_float, f3, f3, C.c_float, f3]
        lib.oracle_fresnel.restype = C.c_float
        lib.oracle_fresnel.argtypes = [f3, f3, C.c_float]
        lib.oracle_refract.restype = None
        lib.oracle_refract.argtypes = [f3, f3, C.c_float, f3]
        lib.oracle_tex_sample.restype = None
        lib.oracle_tex_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, f3]
        lib.oracle_mat_inverse.restype = None
        lib.oracle_mat_inverse.argtypes = [f3, f3]
        lib.oracle_box_pad_factor.restype = C.c_float
        lib.oracle_set_sample_seed.restype = None
        lib.oracle_set_sample_seed.argtypes = [C.c_uint32]
        _lib = lib
    return _lib


def _addr(p):
    """Address of a ctypes pointer / structure / int."""
    if p is None:
        return None
    if isinstance(p, int):
        return p
    if hasattr(p, "contents"):
        return C.addressof(p.contents)
    return C.addressof(p)


def render(desc, opts, camera, width, height, rows=None, threads=1, shadow_any_hit=False, want_keys=True, want_counters=True):
    """Renders rows (start, stop, step) of the frame on the CPU. Returns (rgba, keys, counters)."""
    lib = load()
    rgba = np.zeros((height, width, 4), dtype=np.float32)
    keys = np.full((height, width, 4), -1, dtype=np.int32) if want_keys else None
    cnt = np.zeros(32, dtype=np.uint64) if want_counters else None
    y0, y1, ys = rows if rows else (0, height, 1)
    rc = lib.oracle_render(_addr(desc), _addr(opts), _addr(camera), rgba.ctypes.data, keys.ctypes.data if want_keys else None,
                           cnt.ctypes.data if want_counters else None, y0, y1, ys, threads, 1 if shadow_any_hit else 0)
    if rc != 0:
        raise RuntimeError(f"oracle_render failed ({rc})")
    names = ["primary", "shadow", "secondary", "node", "tri", "sphere", "mesh_entry", "shade_mesh", "shade_sphere"]
    counters = {n: int(cnt[i]) for i, n in enumerate(names)} if want_counters else None
    return rgba, keys, counters
