#!/usr/bin/env python3
"""Development aid: latency of ONE frame rendered as K shards in flight (K streams), each frame synchronised before the
next - does overlapping a frame's own stages across shards shorten the in-order frame?
usage: tools/shard_latency.py [--shards K] [--mode interleaved|rings] [--steps N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import cudaraytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=2)
ap.add_argument("--mode", default="interleaved")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--size", type=int, default=1000000)
a = ap.parse_args()
w, h = 1920, 1080
scene = rt.HostScene("soup", w, h, a.size, 1, 0, tight_bvh=True, device_bvh=True)
with rt.Context(0) as ctx:
    ds = ctx.upload(scene)
    surf, pitch = ctx.surface(w, h)
    K = a.shards
    ctx.set_frames_in_flight(max(K, 1))
    tx, ty = (w + 7) // 8, (h + 7) // 8

    def frame():
        if a.mode == "interleaved" or K == 1:
            for k in range(K):
                ctx.render(ds, scene.opts, scene.camera, surf, pitch, None, None, k, K)
        ctx.sync()

    for _ in range(5):
        frame()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        frame()
    print(f"{K} shard(s), {a.mode}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per frame (each frame synchronised)")
