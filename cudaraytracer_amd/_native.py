"""ctypes binding of librt_gfx950.so, the C-ABI (include/rt_capi.h), and of
librt_host.so, the C++ host classes behind flat entry points
(csrc/host/harness.cpp).

The libraries are never built or substituted here: if one is missing, loading
fails with an error that names the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB (development only) selects another build of the same library, e.g. a tuning variant
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "librt_gfx950.so")
HOST_LIB_PATH = os.path.join(_HERE, "librt_host.so")

RT_COUNTER_NAMES = ["primary", "shadow", "secondary", "node", "tri", "sphere", "mesh_entry", "shade_mesh", "shade_sphere"]
RT_COUNTER_CNT = 32
RT_MAX_DEPTH = 15
RT_STACK_ENTRIES = 64


class RtVertex(C.Structure):
    _fields_ = [("normal", C.c_float * 3), ("tangent", C.c_float * 3), ("bitangent", C.c_float * 3),
                ("texCoord", C.c_float * 2), ("position", C.c_float * 3)]


class RtTriangle(C.Structure):
    _fields_ = [("i", C.c_int32 * 3)]


class RtLinearNode(C.Structure):
    _fields_ = [("bbmin", C.c_float * 3), ("bbmax", C.c_float * 3), ("offset", C.c_int32), ("primCnt", C.c_uint16),
                ("axis", C.c_uint8), ("pad", C.c_uint8)]


class RtTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.POINTER(C.c_ubyte)), ("mode", C.c_int32)]


class RtMaterial(C.Structure):
    _fields_ = [("materialType", C.c_int32), ("ior", C.c_float), ("Kd", C.c_float), ("Ks", C.c_float),
                ("diffuseColor", C.c_float * 3), ("specularExponent", C.c_float)]


class RtLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float * 3), ("power", C.c_float)]


class RtMesh(C.Structure):
    _fields_ = [("triangleCnt", C.c_int32), ("triangles", C.POINTER(RtTriangle)), ("vertexCnt", C.c_int32),
                ("vertices", C.POINTER(RtVertex)), ("diffuse", RtTexture), ("normals", RtTexture), ("specular", RtTexture),
                ("nodeCnt", C.c_int32), ("nodes", C.POINTER(RtLinearNode)), ("transformInverse", C.c_float * 16)]


class RtShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("center", C.c_float * 3), ("radius", C.c_float), ("radius2", C.c_float),
                ("meshIdx", C.c_int32), ("material", RtMaterial), ("bbmin", C.c_float * 3), ("bbmax", C.c_float * 3),
                ("tag", C.c_int32)]


class RtSceneDesc(C.Structure):
    _fields_ = [("lights", C.POINTER(RtLight)), ("lightCnt", C.c_int32), ("shapes", C.POINTER(RtShape)),
                ("shapeCnt", C.c_int32), ("meshes", C.POINTER(RtMesh)), ("meshCnt", C.c_int32),
                ("nodes", C.POINTER(RtLinearNode)), ("nodeCnt", C.c_int32)]


class RtDeviceOpts(C.Structure):
    _fields_ = [("imageAspectRatio", C.c_float), ("scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("backgroundColor", C.c_float * 3), ("shadowBias", C.c_float), ("blockSize", C.c_int32),
                ("maxDepth", C.c_int32), ("samples", C.c_int32), ("gammaCorrection", C.c_uint8),
                ("useSceneBvh", C.c_uint8), ("useMeshBvhs", C.c_uint8), ("pad_", C.c_uint8)]


class RtCamera(C.Structure):
    _fields_ = [("orig", C.c_float * 3), ("camera", C.c_float * 16)]


class RtFrameAux(C.Structure):
    _fields_ = [("hitKeys", C.c_void_p), ("counters", C.c_void_p), ("tileFirst", C.c_int32), ("tileStride", C.c_int32)]


assert C.sizeof(RtVertex) == 56 and C.sizeof(RtLinearNode) == 32 and C.sizeof(RtDeviceOpts) == 48
assert C.sizeof(RtLight) == 28 and C.sizeof(RtMaterial) == 32 and C.sizeof(RtTriangle) == 12

# name -> (restype, argtypes); every symbol include/rt_capi.h declares
C_ABI = {
    "rt_create": (C.c_void_p, [C.c_int]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_sync": (C.c_int, [C.c_void_p]),
    "rt_scene_upload": (C.c_void_p, [C.c_void_p, C.POINTER(RtSceneDesc)]),
    "rt_scene_pack": (C.c_int, [C.c_void_p, C.POINTER(RtSceneDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_blob_free": (None, [C.c_void_p]),
    "rt_scene_adopt_blob": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rt_scene_blob": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_scene_set_lights": (C.c_int, [C.c_void_p, C.POINTER(RtLight), C.c_int]),
    "rt_scene_free": (None, [C.c_void_p]),
    "rt_surface_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_surface_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "rt_surface_free": (None, [C.c_void_p, C.c_void_p]),
    "rt_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rt_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rt_device_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rt_device_free": (None, [C.c_void_p, C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RtDeviceOpts), C.POINTER(RtCamera), C.c_void_p, C.c_size_t,
                            C.c_int, C.POINTER(RtFrameAux)]),
    "rt_timing_reset": (C.c_int, [C.c_void_p]),
    "rt_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "rt_timing_read_stages": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "rt_set_path_budget": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_lds_stack": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_box_test_split": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_device_bvh_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_unlit_shadow_skip": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_mesh_bvh_build": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.c_void_p]),
    "rt_set_frames_in_flight": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_stream_wait_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rt_set_trace_waves": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_tile_order": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_sample_seed": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_set_device_bvh": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_set_upload_refit": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_upload_rebuild": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_bvh_refit": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rt_scene_bvh_refit": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rt_render_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RtDeviceOpts), C.POINTER(RtCamera), C.c_void_p, C.c_size_t,
                                    C.c_int, C.POINTER(RtFrameAux)]),
    "rt_set_tile_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rt_set_tile_cost_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_set_accumulation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]),
    "rt_set_workspace_guards": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_check_workspace_guards": (C.c_int, [C.c_void_p]),
    "rt_tiles_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rt_tiles_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "rt_multi_create": (C.c_void_p, [C.c_int, C.POINTER(C.c_int)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_multi_device_count": (C.c_int, [C.c_void_p]),
    "rt_multi_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rt_multi_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(RtSceneDesc)]),
    "rt_multi_scene_set_lights": (C.c_int, [C.c_void_p, C.POINTER(RtLight), C.c_int]),
    "rt_multi_balance": (C.c_int, [C.c_void_p, C.POINTER(RtDeviceOpts), C.POINTER(RtCamera)]),
    "rt_multi_tile_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rt_multi_set_quantize": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_multi_render": (C.c_int, [C.c_void_p, C.POINTER(RtDeviceOpts), C.POINTER(RtCamera), C.c_void_p, C.c_size_t, C.c_int]),
    "rt_multi_sync": (C.c_int, [C.c_void_p]),
}

HARNESS = {
    "rth_last_error": (C.c_char_p, []),
    "rth_scene_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "rth_scene_free": (None, [C.c_void_p]),
    "rth_scene_desc": (C.POINTER(RtSceneDesc), [C.c_void_p]),
    "rth_scene_opts": (C.POINTER(RtDeviceOpts), [C.c_void_p]),
    "rth_scene_camera": (C.POINTER(RtCamera), [C.c_void_p]),
    "rth_scene_triangles": (C.c_int, [C.c_void_p]),
    "rth_scene_bvh_depth": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rth_scene_rotate_lights": (None, [C.c_void_p, C.c_float]),
    "rth_scene_set_camera": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rth_scene_run_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_void_p]),
    "rth_experiment_open": (C.c_void_p, [C.c_char_p, C.c_char_p]),
    "rth_experiment_free": (None, [C.c_void_p]),
    "rth_experiment_name": (C.c_char_p, [C.c_void_p]),
    "rth_experiment_phase_count": (C.c_int, [C.c_void_p]),
    "rth_experiment_error_count": (C.c_int, [C.c_void_p]),
    "rth_experiment_errors": (C.c_char_p, [C.c_void_p]),
    "rth_experiment_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(RtDeviceOpts), C.POINTER(C.c_int),
                                       C.POINTER(C.c_float), C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_float)]),
    "rth_experiment_scene": (C.c_void_p, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "rth_write_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "rth_floats_to_bytes": (None, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "rth_read_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rth_bvh_triangles": (C.c_int, [C.POINTER(RtVertex), C.c_int, C.POINTER(RtTriangle), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(RtLinearNode), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rth_obj_read": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(RtVertex), C.c_int, C.POINTER(C.c_int), C.c_int,
                               C.c_char_p, C.c_int]),
    "rth_mat_inverse": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rth_selftest": (C.c_int, [C.c_char_p, C.c_int]),
}

_lib = None


class _Libraries:
    """Both libraries behind one namespace: rt_* come from the C-ABI library, rth_* from the host library."""

    def __init__(self, capi, host):
        self.capi = capi
        self.host = host

    def __getattr__(self, name):
        return getattr(self.host if name.startswith("rth_") else self.capi, name)


def load():
    """Loads librt_gfx950.so and librt_host.so (once) and types every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    for path in (LIB_PATH, HOST_LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m cudaraytracer_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
    capi = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)  # the host library's rt_* references resolve to this very build
    host = C.CDLL(HOST_LIB_PATH)
    for lib, table in ((capi, C_ABI), (host, HARNESS)):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
    _lib = _Libraries(capi, host)
    return _lib
