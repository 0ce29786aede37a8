"""rt_render_noalloc against rt_render, through the C-ABI: same pixels, same hit keys, same counters, on two calls in
a row (the second with a smaller frame, so that the shared static scratch array still holds the first call's
contents), then a frame rendered shard by shard and a frame that does not fit."""
import numpy as np
import pytest

import cudaraytracer_amd as rt

pytestmark = pytest.mark.gpu


def _frame(ctx, scene, dscene, noalloc, w, h, shard=(0, 1)):
    surf, pitch = ctx.surface(w, h)
    keys = ctx.alloc(w * h * 16)
    cnt = ctx.alloc(rt.RT_COUNTER_CNT * 8)
    try:
        render = ctx.render_noalloc if noalloc else ctx.render
        for first in range(shard[0], shard[1]):
            render(dscene, scene.opts, scene.camera, surf, pitch, keys, cnt, first, shard[1])
        ctx.sync()
        raw = ctx.download(cnt, (rt.RT_COUNTER_CNT,), np.uint64)
        return ctx.download_surface(surf, pitch, w, h), ctx.download(keys, (h, w, 4), np.int32), raw[:16].copy()
    finally:
        for p in (surf, keys, cnt):
            ctx.free(p)


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def test_noalloc_matches_rt_render_on_consecutive_calls():
    ctx = rt.Context(0)
    try:
        # mixed: spheres, mirror + glass (child rays, path-node levels), textured meshes, 3 lights, depth 3
        for (w, h) in ((160, 120), (96, 64)):  # the second call is smaller and runs on the first one's leftovers
            scene = rt.HostScene("mixed", w, h, 0, 3, 3)
            dscene = ctx.upload(scene)
            want = _frame(ctx, scene, dscene, False, w, h)
            got = _frame(ctx, scene, dscene, True, w, h)
            assert _same(got[0], want[0]), f"{w}x{h}: pixels differ"
            assert (got[1] == want[1]).all(), f"{w}x{h}: hit keys differ"
            assert (got[2] == want[2]).all(), f"{w}x{h}: counters differ {got[2]} {want[2]}"
            ctx.free_scene(dscene)
    finally:
        ctx.close()


def test_noalloc_calls_leave_free_device_memory_unchanged():
    """The contract "no hipMalloc / hipFree": after a warm-up call (module load, first-use set-up), free device memory as
    hipMemGetInfo reports it stays the same over a series of rt_render_noalloc calls of different sizes, with and
    without counters, with a tile list set beforehand."""
    import torch

    ctx = rt.Context(0)
    try:
        scenes = [rt.HostScene("mixed", 160, 120, 0, 3, 3), rt.HostScene("cubes", 96, 64, 4, 2, 0)]
        dscenes = [ctx.upload(s) for s in scenes]
        bufs = [(ctx.surface(s.width, s.height), ctx.alloc(s.width * s.height * 16), ctx.alloc(rt.RT_COUNTER_CNT * 8)) for s in scenes]
        ctx.set_tile_list(list(range(0, 96, 3)))
        for s, d, ((surf, pitch), keys, cnt) in zip(scenes, dscenes, bufs):  # warm-up
            ctx.render_noalloc(d, s.opts, s.camera, surf, pitch, keys, cnt)
        ctx.sync()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        for rep in range(3):
            for s, d, ((surf, pitch), keys, cnt) in zip(scenes, dscenes, bufs):
                ctx.render_noalloc(d, s.opts, s.camera, surf, pitch, keys if rep % 2 else None, cnt if rep == 2 else None)
        ctx.sync()
        assert torch.cuda.mem_get_info(0)[0] == free_before
        # the allocating entry point does move it on a fresh context (first workspace): the probe can see an allocation
        ctx.set_tile_list(None)
        ctx.render(dscenes[0], scenes[0].opts, scenes[0].camera, bufs[0][0][0], bufs[0][0][1])
        ctx.sync()
        assert torch.cuda.mem_get_info(0)[0] < free_before
    finally:
        ctx.close()


def test_noalloc_tile_shards_and_limit():
    ctx = rt.Context(0)
    try:
        w, h = 1280, 720  # 921600 pixels x 4 lights: more than one call's scratch; four shards of it fit
        scene = rt.HostScene("cubes", w, h, 6, 4, 0)
        dscene = ctx.upload(scene)
        surf, pitch = ctx.surface(w, h)
        with pytest.raises(rt.RtError, match="scratch"):
            ctx.render_noalloc(dscene, scene.opts, scene.camera, surf, pitch)
        ctx.free(surf)
        want = _frame(ctx, scene, dscene, False, w, h)
        got = _frame(ctx, scene, dscene, True, w, h, shard=(0, 4))
        assert _same(got[0], want[0]) and (got[1] == want[1]).all()
        assert (got[2] == want[2]).all()
    finally:
        ctx.close()
