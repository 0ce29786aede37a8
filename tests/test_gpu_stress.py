"""Frames in flight under changing conditions: scenes with and without reflection levels, frame sizes that make a
slot's workspace grow, tile shards (more shardings than the tile-order cache holds), buffers allocated while frames
run, lights changed between frames.  Every frame must equal the same frame rendered strictly in order."""
import numpy as np
import pytest

import cudaraytracer_amd as rt

pytestmark = pytest.mark.gpu

SCENES = [("mixed", 0, 3, 3), ("soup", 20000, 2, 0), ("cubes", 6, 4, 2)]
SIZES = [(64, 48), (200, 120), (96, 160)]


def test_random_sequence_of_frames_in_flight(gpu_ctx):
    rng = np.random.default_rng(2024)
    scenes, dscenes, expect = {}, {}, {}
    for si, (name, size, lights, depth) in enumerate(SCENES):
        for zi, (w, h) in enumerate(SIZES):
            s = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=True)
            scenes[si, zi], dscenes[si, zi] = s, gpu_ctx.upload(s)
    surfs = []
    gpu_ctx.set_path_budget(16)   # a shard can consist of glass tiles only
    try:
        jobs = []
        for _ in range(90):
            si, zi = int(rng.integers(len(SCENES))), int(rng.integers(len(SIZES)))
            stride = int(rng.integers(1, 40))          # up to 39 x 3 sizes shardings: the cache of 64 tables turns over
            first = int(rng.integers(stride))
            jobs.append((si, zi, first, stride))
        # references, strictly in order
        for job in set(jobs):
            si, zi, first, stride = job
            w, h = SIZES[zi]
            surf, pitch = gpu_ctx.surface(w, h)
            gpu_ctx.render(dscenes[si, zi], scenes[si, zi].opts, scenes[si, zi].camera, surf, pitch, None, None, first, stride)
            gpu_ctx.sync()
            expect[job] = gpu_ctx.download_surface(surf, pitch, w, h)
            gpu_ctx.free(surf)
        gpu_ctx.set_frames_in_flight(5)
        for job in jobs:
            si, zi, first, stride = job
            w, h = SIZES[zi]
            surfs.append(gpu_ctx.surface(w, h))         # allocated (and filled) while earlier frames are still running
            gpu_ctx.render(dscenes[si, zi], scenes[si, zi].opts, scenes[si, zi].camera, surfs[-1][0], surfs[-1][1], None, None, first, stride)
        gpu_ctx.sync()
        for i, job in enumerate(jobs):
            w, h = SIZES[job[1]]
            got = gpu_ctx.download_surface(surfs[i][0], surfs[i][1], w, h)
            assert np.array_equal(got, expect[job], equal_nan=True), (i, job)
    finally:
        gpu_ctx.set_frames_in_flight(1)
        gpu_ctx.set_path_budget(3)
        for sf in surfs:
            gpu_ctx.free(sf[0])
        for d in dscenes.values():
            gpu_ctx.free_scene(d)


def test_lights_change_between_frames_in_flight(gpu_ctx, oracle):
    """rt_scene_set_lights waits for the frames in flight: a frame keeps the lights it was issued with."""
    w, h = 128, 96
    scene = rt.HostScene("mixed", w, h, 0, 3, 2)
    dscene = gpu_ctx.upload(scene)
    surfs = [gpu_ctx.surface(w, h) for _ in range(6)]
    refs = []
    try:
        gpu_ctx.set_frames_in_flight(3)
        for i in range(6):
            refs.append(oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8)[0])
            gpu_ctx.render(dscene, scene.opts, scene.camera, surfs[i][0], surfs[i][1])
            scene.rotate_lights(0.5)
            gpu_ctx.set_lights(dscene, scene.desc)
        gpu_ctx.sync()
        from parity import assert_parity

        for i in range(6):
            assert_parity(gpu_ctx.download_surface(surfs[i][0], surfs[i][1], w, h), None, refs[i], None)
        assert not np.array_equal(refs[0], refs[5])
    finally:
        gpu_ctx.set_frames_in_flight(1)
        for sf in surfs:
            gpu_ctx.free(sf[0])
        gpu_ctx.free_scene(dscene)
