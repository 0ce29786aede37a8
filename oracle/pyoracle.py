"""ctypes loader of oracle/liboracle.so - TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the cudaraytracer_amd package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
# the same source at -O3 -march=x86-64-v3 with contraction: NOT the parity reference (its roundings differ), only a
# faster CPU figure for the bench's baseline
O3_LIB_PATH = os.path.join(_HERE, "liboracle_o3.so")
_libs = {}


def o3_supported():
    """True if this CPU has what an x86-64-v3 build needs and the -O3 library exists."""
    if not os.path.exists(O3_LIB_PATH):
        return False
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((line.split(":", 1)[1].split() for line in f if line.startswith("flags")), [])
    except OSError:
        return False
    return {"avx", "avx2", "fma", "bmi1", "bmi2", "f16c", "abm", "movbe"} <= set(flags)


def load(o3=False):
    path = O3_LIB_PATH if o3 else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -m cudaraytracer_amd.build`")
        if o3 and not o3_supported():
            raise RuntimeError("this CPU cannot run the x86-64-v3 build of the oracle")
        lib = C.CDLL(path)
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
        _libs[path] = lib
    return _libs[path]


def _addr(p):
    """Address of a ctypes pointer / structure / int."""
    if p is None:
        return None
    if isinstance(p, int):
        return p
    if hasattr(p, "contents"):
        return C.addressof(p.contents)
    return C.addressof(p)


def render(desc, opts, camera, width, height, rows=None, threads=1, shadow_any_hit=False, want_keys=True, want_counters=True,
           o3=False):
    """Renders rows (start, stop, step) of the frame on the CPU. Returns (rgba, keys, counters)."""
    lib = load(o3)
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
