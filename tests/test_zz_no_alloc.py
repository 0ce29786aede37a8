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
