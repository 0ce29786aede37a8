"""Progressive accumulation over frames on the device (rt_set_accumulation): N calls of one sample add up, bit for bit,
to one call of N samples - and that one is held to the oracle's N-sample frame (same jitter hash, samples summed in
order, cudaTracer/Main.cu:66-98)."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity

pytestmark = pytest.mark.gpu


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("name,w,h,total,per_call,gamma", [("mixed", 96, 64, 8, 1, 0), ("mixed", 96, 64, 8, 1, 1), ("cubes", 64, 48, 12, 3, 0),
                                                            ("soup", 101, 50, 5, 1, 0)])
def test_accumulated_frames_equal_one_multi_sample_frame(gpu_ctx, oracle, name, w, h, total, per_call, gamma):
    scene = rt.HostScene(name, w, h, 20000 if name == "soup" else 6, 3, 2)
    o = scene.opts.contents
    o.gammaCorrection = gamma
    gpu_ctx.set_sample_seed(777)
    oracle.load().oracle_set_sample_seed(777)
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(w, h)
    acc, apitch = gpu_ctx.surface(w, h)  # starts as garbage (0x01 bytes): samplesSoFar = 0 must ignore it
    keys = gpu_ctx.alloc(w * h * 16)
    try:
        # one call with all the samples: the frame to match, itself checked against the oracle
        o.samples = total
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys)
        gpu_ctx.sync()
        whole = gpu_ctx.download_surface(surf, pitch, w, h)
        whole_keys = gpu_ctx.download(keys, (h, w, 4), np.int32)
        ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
        assert_parity(whole, whole_keys, ref_rgba, ref_keys)
        # the same samples over several calls
        o.samples = per_call
        gpu_ctx.set_accumulation(acc, apitch, 0)
        partial = None
        for k in range(total // per_call):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys)
            if k == 0:
                gpu_ctx.sync()
                partial = gpu_ctx.download_surface(surf, pitch, w, h)
        gpu_ctx.sync()
        got = gpu_ctx.download_surface(surf, pitch, w, h)
        got_keys = gpu_ctx.download(keys, (h, w, 4), np.int32)
        sums = gpu_ctx.download_surface(acc, apitch, w, h)
        if gamma == 0:
            assert _same(got, whole)
        else:  # powf may differ in the last place between two evaluations only if its argument does: it does not
            assert _same(got, whole)
        assert np.array_equal(got_keys, whole_keys)  # the keys are sample 0's, written by the first call only
        assert (sums[..., 3] == total).all()
        assert not _same(partial, got)  # the first call alone is a different (noisier) frame
        # going on after a pause: tell the library how many samples the surface holds
        gpu_ctx.set_accumulation(None)
        gpu_ctx.set_accumulation(acc, apitch, total)
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        gpu_ctx.sync()
        assert (gpu_ctx.download_surface(acc, apitch, w, h)[..., 3] == total + per_call).all()
    finally:
        gpu_ctx.set_accumulation(None)
        gpu_ctx.set_sample_seed(1234)
        oracle.load().oracle_set_sample_seed(1234)
        o.samples = 1
        for p in (surf, acc, keys):
            gpu_ctx.free(p)
        gpu_ctx.free_scene(dscene)


def test_accumulation_needs_frames_in_order(gpu_ctx):
    scene = rt.HostScene("cubes", 64, 48, 3, 1, 0)
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(64, 48)
    acc, apitch = gpu_ctx.surface(64, 48)
    try:
        gpu_ctx.set_frames_in_flight(2)
        gpu_ctx.set_accumulation(acc, apitch, 0)
        with pytest.raises(rt.RtError, match="in order"):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
    finally:
        gpu_ctx.set_accumulation(None)
        gpu_ctx.set_frames_in_flight(1)
        gpu_ctx.free(surf)
        gpu_ctx.free(acc)
        gpu_ctx.free_scene(dscene)


@pytest.mark.parametrize("noalloc", [False, True])
def test_accumulation_of_frames_rendered_shard_by_shard(gpu_ctx, noalloc):
    """N one-sample frames, each rendered as four tile shards on ONE context, equal one N-sample frame bit for bit:
    every pixel continues from its own sample count, whichever call renders it and in whatever order the shards come."""
    w, h, total, shards = 101, 67, 6, 4
    scene = rt.HostScene("mixed", w, h, 0, 3, 2)
    o = scene.opts.contents
    gpu_ctx.set_sample_seed(4242)
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(w, h)
    acc, apitch = gpu_ctx.surface(w, h)  # 0x01 bytes: a new accumulation must not read them
    keys = gpu_ctx.alloc(w * h * 16)
    render = gpu_ctx.render_noalloc if noalloc else gpu_ctx.render
    try:
        o.samples = total
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys)
        gpu_ctx.sync()
        whole = gpu_ctx.download_surface(surf, pitch, w, h)
        whole_keys = gpu_ctx.download(keys, (h, w, 4), np.int32)
        o.samples = 1
        gpu_ctx.zero(keys, w * h * 16)
        gpu_ctx.set_accumulation(acc, apitch, 0)
        for frame in range(total):
            order = list(range(shards)) if frame % 2 == 0 else list(reversed(range(shards)))
            for first in order:
                render(dscene, scene.opts, scene.camera, surf, pitch, keys, None, first, shards)
        gpu_ctx.sync()
        assert _same(gpu_ctx.download_surface(surf, pitch, w, h), whole)
        assert np.array_equal(gpu_ctx.download(keys, (h, w, 4), np.int32), whole_keys)
        assert (gpu_ctx.download_surface(acc, apitch, w, h)[..., 3] == total).all()
    finally:
        gpu_ctx.set_accumulation(None)
        gpu_ctx.set_sample_seed(1234)
        o.samples = 1
        for p in (surf, acc, keys):
            gpu_ctx.free(p)
        gpu_ctx.free_scene(dscene)
