"""Thin Python mirror of the host API: HostScene (a scene built by the C++
RayTracer / DeviceMesh / BVH::BvhFactory classes) and Context (one GPU, all
calls going through the C-ABI)."""
import ctypes as C

import numpy as np

from . import _native as N


class RtError(RuntimeError):
    pass


class HostScene:
    """A benchmark scene assembled on the host (csrc/host/harness.cpp).

    name: cubes | soup | heightfield | mixed | barrel
    tight_bvh: start BVH box unions from a true empty box instead of the
    reference's FLT_MIN-clamped one (same hits except ties, far fewer tests).
    sah_bvh: cut mesh BVHs by the binned surface-area heuristic instead of the
    reference's median cut (host build; same hits except ties, fewer tests);
    leaf_threshold: most triangles in a leaf (0: 10 for the median cut, 8 for
    the surface-area cut).
    """

    def __init__(self, name, width, height, size=0, lights=0, max_depth=0, asset_dir=None, tight_bvh=False, device_bvh=False,
                 sah_bvh=False, leaf_threshold=0, _handle=None):
        self.lib = N.load()
        if _handle is not None:  # a scene some other harness call built (Experiment.scene)
            self.h = _handle
            self.name = name
            return
        self.h = self.lib.rth_scene_create(name.encode(), width, height, size, lights, max_depth,
                                           asset_dir.encode() if asset_dir else None, (1 if tight_bvh else 0) | (2 if device_bvh else 0) | (8 if sah_bvh else 0) | ((leaf_threshold & 0xff) << 8))
        if not self.h:
            raise RtError(self.lib.rth_last_error().decode())
        self.name = name

    def close(self):
        if self.h:
            self.lib.rth_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        return self.lib.rth_scene_desc(self.h)

    @property
    def opts(self):
        """Pointer to the scene's DeviceOpts (mutable: opts.contents.maxDepth = 2)."""
        return self.lib.rth_scene_opts(self.h)

    @property
    def camera(self):
        return self.lib.rth_scene_camera(self.h)

    @property
    def width(self):
        return self.opts.contents.width

    @property
    def height(self):
        return self.opts.contents.height

    @property
    def triangles(self):
        return self.lib.rth_scene_triangles(self.h)

    def bvh_depth(self):
        a, b = C.c_int(), C.c_int()
        self.lib.rth_scene_bvh_depth(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def run_frames(self, frames, dt, png_path=None, device=0):
        """The C++ RayTracer driver end to end: upload, `frames` x (rotate lights by dt, render), dump; returns the last frame."""
        rgba = np.zeros((self.height, self.width, 4), dtype=np.float32)
        rc = self.lib.rth_scene_run_frames(self.h, device, frames, dt, png_path.encode() if png_path else None, rgba.ctypes.data)
        if rc != 0:
            raise RtError(self.lib.rth_last_error().decode())
        return rgba

    def rotate_lights(self, dt):
        self.lib.rth_scene_rotate_lights(self.h, dt)

    def set_camera(self, pos, direction):
        p = (C.c_float * 3)(*pos)
        d = (C.c_float * 3)(*direction)
        self.lib.rth_scene_set_camera(self.h, p, d)


class Experiment:
    """An experiment description in the reference's JSON format (csrc/host/ExperimentReader.h,
    cudaTracer/ExperimentReader.cpp): a name and up to five phases, each a set of option overrides on the defaults.

    Experiment(path) reads a file, Experiment(text=...) a string.  `errors` lists the messages the reader
    collected (wrong types, missing names, bad JSON); like the reference, reading carries on past them.
    """

    def __init__(self, path=None, text=None):
        self.lib = N.load()
        if (path is None) == (text is None):
            raise ValueError("Experiment: give a path or a text")
        self.h = self.lib.rth_experiment_open(str(path).encode() if path is not None else None,
                                              text.encode() if text is not None else None)
        if not self.h:
            raise RtError(self.lib.rth_last_error().decode())
        self.name = self.lib.rth_experiment_name(self.h).decode()
        raw = self.lib.rth_experiment_errors(self.h).decode()
        self.errors = [m for m in raw.split("\n\n") if m]
        self.phases = [self._phase(i) for i in range(self.lib.rth_experiment_phase_count(self.h))]

    def _phase(self, idx):
        name, folder, fname = (C.create_string_buffer(512) for _ in range(3))
        dev = N.RtDeviceOpts()
        lights, fov = C.c_int(), C.c_float()
        transform = (C.c_float * 16)()
        rc = self.lib.rth_experiment_phase(self.h, idx, name, 512, C.byref(dev), C.byref(lights), C.byref(fov), folder, 512, fname, 512,
                                           transform)
        if rc != 0:
            raise RtError(self.lib.rth_last_error().decode())
        return {
            "name": name.value.decode(),
            "device": dev,
            "lightCnt": lights.value,
            "fov": fov.value,
            "model": {"folder": folder.value.decode(), "fname": fname.value.decode(),
                      "transform": np.array(transform, dtype=np.float32).reshape(4, 4)},
        }

    def scene(self, idx, asset_dir=None, tight_bvh=False, device_bvh=False, derive_view=False, sah_bvh=False):
        """The scene of phase idx, assembled like RayTracer::initScene (cudaTracer/RayTracer.cpp:50-78).
        derive_view: recompute scale / aspect ratio from the phase's fov, width, height (the reference keeps the defaults').
        sah_bvh: mesh BVHs cut by the binned surface-area heuristic instead of the reference's median cut."""
        flags = (1 if tight_bvh else 0) | (2 if device_bvh else 0) | (4 if derive_view else 0) | (8 if sah_bvh else 0)
        h = self.lib.rth_experiment_scene(self.h, idx, asset_dir.encode() if asset_dir else None, flags)
        if not h:
            raise RtError(self.lib.rth_last_error().decode())
        return HostScene(self.phases[idx]["name"], 0, 0, _handle=h)

    def close(self):
        if self.h:
            self.lib.rth_experiment_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tiles_of_rank(width, height, rank, world, tile=8):
    """Tile ids (row-major over 8x8 tiles) rendered by `rank` of `world`: t = rank, rank+world, ..."""
    tx = (width + tile - 1) // tile
    ty = (height + tile - 1) // tile
    return np.arange(rank, tx * ty, world, dtype=np.int64), tx, ty


def pack_scene(lib, desc, ctx=None):
    """rt_scene_pack: flattens a scene description into the relocatable device blob on the host (needs no GPU)."""
    p, n = C.c_void_p(), C.c_size_t()
    rc = lib.rt_scene_pack(ctx, desc, C.byref(p), C.byref(n))
    if rc != 0:
        raise RtError(f"rt_scene_pack failed ({rc}): {lib.rt_last_error(ctx).decode()}")
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(n.value,)).copy()
    finally:
        lib.rt_blob_free(p)


class Context:
    """One GPU. Owns device-side scenes, surfaces and aux buffers via the C-ABI."""

    def __init__(self, device=0):
        self.lib = N.load()
        self.ctx = self.lib.rt_create(device)
        if not self.ctx:
            raise RtError("rt_create: " + self.lib.rt_last_error(None).decode())
        self._allocs = []
        self._scenes = []

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {self.lib.rt_last_error(self.ctx).decode()}")

    def close(self):
        if not self.ctx:
            return
        for s in self._scenes:
            self.lib.rt_scene_free(s)
        for p in self._allocs:
            self.lib.rt_device_free(self.ctx, p)
        self._scenes, self._allocs = [], []
        self.lib.rt_destroy(self.ctx)
        self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene -------------------------------------------------------------
    def upload(self, scene_or_desc):
        desc = scene_or_desc.desc if isinstance(scene_or_desc, HostScene) else scene_or_desc
        s = self.lib.rt_scene_upload(self.ctx, desc)
        if not s:
            raise RtError("rt_scene_upload: " + self.lib.rt_last_error(self.ctx).decode())
        self._scenes.append(s)
        return s

    def adopt_blob(self, dev_ptr, nbytes):
        s = self.lib.rt_scene_adopt_blob(self.ctx, dev_ptr, nbytes)
        if not s:
            raise RtError("rt_scene_adopt_blob: " + self.lib.rt_last_error(self.ctx).decode())
        self._scenes.append(s)
        return s

    def scene_blob(self, scene):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rt_scene_blob(scene, C.byref(p), C.byref(n)), "rt_scene_blob")
        return p.value, n.value

    def pack(self, desc):
        """Host-side flattening only: returns the blob as a numpy uint8 array."""
        return pack_scene(self.lib, desc, self.ctx)

    def set_lights(self, scene, desc):
        d = desc.contents if hasattr(desc, "contents") else desc
        self._check(self.lib.rt_scene_set_lights(scene, d.lights, d.lightCnt), "rt_scene_set_lights")

    def free_scene(self, scene):
        self._scenes.remove(scene)
        self.lib.rt_scene_free(scene)

    # -- buffers -----------------------------------------------------------
    def surface(self, width, height):
        p, pitch = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rt_surface_create(self.ctx, width, height, C.byref(p), C.byref(pitch)), "rt_surface_create")
        self._allocs.append(p.value)
        return p.value, pitch.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.rt_device_alloc(self.ctx, nbytes, C.byref(p)), "rt_device_alloc")
        self._allocs.append(p.value)
        return p.value

    def free(self, ptr):
        self._allocs.remove(ptr)
        self.lib.rt_device_free(self.ctx, ptr)

    def zero(self, ptr, nbytes):
        self._check(self.lib.rt_device_zero(self.ctx, ptr, nbytes), "rt_device_zero")

    def download_surface(self, surf, pitch, width, height):
        out = np.empty((height, width, 4), dtype=np.float32)
        self._check(self.lib.rt_surface_download(self.ctx, surf, pitch, width, height, out.ctypes.data), "rt_surface_download")
        return out

    def download(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.rt_device_download(self.ctx, ptr, out.nbytes, out.ctypes.data), "rt_device_download")
        return out

    # -- rendering ---------------------------------------------------------
    def render(self, scene, opts, camera, surf, pitch, hit_keys=None, counters=None, tile_first=0, tile_stride=1):
        aux = N.RtFrameAux(hit_keys, counters, tile_first, tile_stride)
        o = opts.contents if hasattr(opts, "contents") else opts
        self._check(self.lib.rt_render(self.ctx, scene, C.byref(o), camera, surf, pitch, o.blockSize, C.byref(aux)), "rt_render")

    def render_noalloc(self, scene, opts, camera, surf, pitch, hit_keys=None, counters=None, tile_first=0, tile_stride=1):
        """rt_render_noalloc: same frame, workspace from the library's static device scratch array, no allocation."""
        aux = N.RtFrameAux(hit_keys, counters, tile_first, tile_stride)
        o = opts.contents if hasattr(opts, "contents") else opts
        self._check(self.lib.rt_render_noalloc(self.ctx, scene, C.byref(o), camera, surf, pitch, o.blockSize, C.byref(aux)), "rt_render_noalloc")

    def tiles_pack(self, surf, pitch, width, height, dev_tiles, count, dev_out, quantize=False):
        """rt_tiles_pack: the pixels of `count` 8x8 tiles (int32 tile indices in device memory) out of a float4 surface."""
        self._check(self.lib.rt_tiles_pack(self.ctx, surf, pitch, width, height, dev_tiles, count, dev_out, 1 if quantize else 0), "rt_tiles_pack")

    def tiles_scatter(self, dev_in, dev_tiles, count, surf, pitch, width, height, quantized=False):
        """rt_tiles_scatter: packed tiles to their places in a surface (float4, or 8-bit RGBA when quantized)."""
        self._check(self.lib.rt_tiles_scatter(self.ctx, dev_in, dev_tiles, count, surf, pitch, width, height, 1 if quantized else 0), "rt_tiles_scatter")

    def sync(self):
        self._check(self.lib.rt_sync(self.ctx), "rt_sync")

    def set_workspace_guards(self, nbytes):
        self._check(self.lib.rt_set_workspace_guards(self.ctx, nbytes), "rt_set_workspace_guards")

    def check_workspace_guards(self):
        self._check(self.lib.rt_check_workspace_guards(self.ctx), "rt_check_workspace_guards")

    def set_stream(self, stream_ptr):
        self._check(self.lib.rt_set_stream(self.ctx, stream_ptr), "rt_set_stream")

    def timing_reset(self):
        self._check(self.lib.rt_timing_reset(self.ctx), "rt_timing_reset")

    def timing_read(self):
        ms, n = C.c_double(), C.c_int()
        self._check(self.lib.rt_timing_read(self.ctx, C.byref(ms), C.byref(n)), "rt_timing_read")
        return ms.value, n.value

    def timing_read_stages(self):
        """(total ms, [primary trace, shade, shadow trace, rest] ms, launches) since the last reset."""
        ms, n = C.c_double(), C.c_int()
        st = (C.c_double * 4)()
        self._check(self.lib.rt_timing_read_stages(self.ctx, C.byref(ms), st, C.byref(n)), "rt_timing_read_stages")
        return ms.value, list(st), n.value

    def set_path_budget(self, nodes_per_pixel):
        self._check(self.lib.rt_set_path_budget(self.ctx, nodes_per_pixel), "rt_set_path_budget")

    def set_trace_waves(self, waves_per_cu):
        self._check(self.lib.rt_set_trace_waves(self.ctx, waves_per_cu), "rt_set_trace_waves")

    def set_device_bvh(self, threshold=10, tight_bounds=True):
        """Parameters of the on-device BVH build used for meshes that come without nodes."""
        self._check(self.lib.rt_set_device_bvh(self.ctx, threshold, 1 if tight_bounds else 0), "rt_set_device_bvh")

    def set_upload_refit(self, on=True):
        """Recompute the boxes of the trees a scene arrives with at upload (same topology, true bounds)."""
        self._check(self.lib.rt_set_upload_refit(self.ctx, 1 if on else 0), "rt_set_upload_refit")

    def set_upload_rebuild(self, on=True):
        """Build every mesh BVH on the device at upload, whether or not the mesh comes with nodes (rt_set_device_bvh's rule)."""
        self._check(self.lib.rt_set_upload_rebuild(self.ctx, 1 if on else 0), "rt_set_upload_rebuild")

    def set_sample_seed(self, seed):
        self._check(self.lib.rt_set_sample_seed(self.ctx, seed), "rt_set_sample_seed")

    def set_tile_order(self, mode):
        """0 = row-major, 1 = centre-out (default)."""
        self._check(self.lib.rt_set_tile_order(self.ctx, mode), "rt_set_tile_order")

    def set_frames_in_flight(self, frames):
        """1 = frames run in order on the context's stream; 2..8 = consecutive frames overlap on internal streams."""
        self._check(self.lib.rt_set_frames_in_flight(self.ctx, frames), "rt_set_frames_in_flight")

    def set_accumulation(self, dev_sum, pitch=0, samples_so_far=0):
        """Running-sum surface for progressive accumulation over frames (None ends it)."""
        self._check(self.lib.rt_set_accumulation(self.ctx, dev_sum, pitch, samples_so_far), "rt_set_accumulation")

    def set_tile_list(self, tiles):
        """The tiles (frame tile indices, in queue order) of the following render calls; None returns to tileFirst / tileStride."""
        if tiles is None or len(tiles) == 0:
            self._check(self.lib.rt_set_tile_list(self.ctx, None, 0), "rt_set_tile_list")
            return
        a = np.ascontiguousarray(np.asarray(tiles, dtype=np.int32))
        self._check(self.lib.rt_set_tile_list(self.ctx, a.ctypes.data, int(a.size)), "rt_set_tile_list")

    def set_tile_cost_buffer(self, dev_ptr):
        self._check(self.lib.rt_set_tile_cost_buffer(self.ctx, dev_ptr), "rt_set_tile_cost_buffer")

    def stream_wait_frame(self, stream_ptr=None, frames_ago=0):
        self._check(self.lib.rt_stream_wait_frame(self.ctx, stream_ptr, frames_ago), "rt_stream_wait_frame")

    def set_box_test_split(self, min_rays=100000):
        """Level-0 traces of at least this many primary rays run as two launches (min / max box test, then the literal one)."""
        self._check(self.lib.rt_set_box_test_split(self.ctx, min_rays), "rt_set_box_test_split")

    def set_device_bvh_rule(self, sah=False):
        """rt_set_device_bvh_rule: meshes built on the device are cut at the median (the reference's rule) or by the binned
        surface-area heuristic (the tree of the host builder with that rule, decided on the device)."""
        self._check(self.lib.rt_set_device_bvh_rule(self.ctx, 1 if sah else 0), "rt_set_device_bvh_rule")

    def set_unlit_shadow_skip(self, on=True):
        """rt_set_unlit_shadow_skip: shadow rays towards lights behind the surface (lambert == 0) are not walked; same pixels."""
        self._check(self.lib.rt_set_unlit_shadow_skip(self.ctx, 1 if on else 0), "rt_set_unlit_shadow_skip")

    def set_lds_stack(self, entries):
        self._check(self.lib.rt_set_lds_stack(self.ctx, entries), "rt_set_lds_stack")

    def render_frame(self, host_scene, scene=None, want_keys=False, want_counters=False):
        """Convenience: full frame of a HostScene -> (rgba[h,w,4], keys or None, counters dict or None)."""
        own = scene is None
        if own:
            scene = self.upload(host_scene)
        w, h = host_scene.width, host_scene.height
        surf, pitch = self.surface(w, h)
        keys = self.alloc(w * h * 16) if want_keys else None
        cnt = self.alloc(N.RT_COUNTER_CNT * 8) if want_counters else None
        try:
            self.render(scene, host_scene.opts, host_scene.camera, surf, pitch, keys, cnt)
            self.sync()
            rgba = self.download_surface(surf, pitch, w, h)
            k = self.download(keys, (h, w, 4), np.int32) if want_keys else None
            c = None
            if want_counters:
                raw = self.download(cnt, (N.RT_COUNTER_CNT,), np.uint64)
                c = {name: int(raw[i]) for i, name in enumerate(N.RT_COUNTER_NAMES)}
            return rgba, k, c
        finally:
            self.free(surf)
            if keys:
                self.free(keys)
            if cnt:
                self.free(cnt)
            if own:
                self.free_scene(scene)


class MultiContext:
    """Several GPUs driven from this one process (rt_multi_* of the C-ABI): the scene uploaded once and copied to the
    peers, every device rendering its own tiles of the same frame, the tiles gathered into a surface on devices[0].
    A device may be named more than once (several contexts on one GPU)."""

    def __init__(self, devices):
        self.lib = N.load()
        arr = (C.c_int * len(devices))(*devices)
        self.m = self.lib.rt_multi_create(len(devices), arr)
        if not self.m:
            raise RtError("rt_multi_create: " + self.lib.rt_multi_last_error(None).decode())
        self.ndev = len(devices)
        self._allocs = []

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {self.lib.rt_multi_last_error(self.m).decode()}")

    def _check0(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {self.lib.rt_last_error(self.raw_context(0)).decode()}")

    def raw_context(self, i):
        """The i-th device's rt_ctx* (for the per-context knobs, e.g. lib.rt_set_upload_refit(raw_context(0), 1))."""
        return self.lib.rt_multi_context(self.m, i)

    def close(self):
        if not self.m:
            return
        for p in self._allocs:
            self.lib.rt_device_free(self.raw_context(0), p)
        self._allocs = []
        self.lib.rt_multi_destroy(self.m)
        self.m = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, scene_or_desc):
        desc = scene_or_desc.desc if isinstance(scene_or_desc, HostScene) else scene_or_desc
        self._check(self.lib.rt_multi_scene_upload(self.m, desc), "rt_multi_scene_upload")

    def set_lights(self, desc):
        d = desc.contents if hasattr(desc, "contents") else desc
        self._check(self.lib.rt_multi_scene_set_lights(self.m, d.lights, d.lightCnt), "rt_multi_scene_set_lights")

    def balance(self, opts, camera):
        o = opts.contents if hasattr(opts, "contents") else opts
        self._check(self.lib.rt_multi_balance(self.m, C.byref(o), camera), "rt_multi_balance")

    def tile_counts(self):
        out = (C.c_int * self.ndev)()
        self._check(self.lib.rt_multi_tile_counts(self.m, out), "rt_multi_tile_counts")
        return list(out)

    def set_quantize(self, on):
        self._check(self.lib.rt_multi_set_quantize(self.m, 1 if on else 0), "rt_multi_set_quantize")

    def surface(self, width, height, bytes_per_pixel=16):
        """A surface on devices[0]: float4 pixels (16) or 8-bit RGBA (4).  Returns (pointer, pitch)."""
        p = C.c_void_p()
        pitch = (width * bytes_per_pixel + 255) // 256 * 256
        self._check0(self.lib.rt_device_alloc(self.raw_context(0), pitch * height, C.byref(p)), "rt_device_alloc")
        self._allocs.append(p.value)
        return p.value, pitch

    def download(self, ptr, pitch, width, height, bytes_per_pixel=16):
        raw = np.empty((height, pitch), dtype=np.uint8)
        self._check0(self.lib.rt_device_download(self.raw_context(0), ptr, raw.nbytes, raw.ctypes.data), "rt_device_download")
        row = raw[:, : width * bytes_per_pixel]
        if bytes_per_pixel == 16:
            return np.ascontiguousarray(row).view(np.float32).reshape(height, width, 4)
        return np.ascontiguousarray(row).reshape(height, width, 4)

    def render(self, opts, camera, surf, pitch):
        o = opts.contents if hasattr(opts, "contents") else opts
        self._check(self.lib.rt_multi_render(self.m, C.byref(o), camera, surf, pitch, o.blockSize), "rt_multi_render")

    def sync(self):
        self._check(self.lib.rt_multi_sync(self.m), "rt_multi_sync")
