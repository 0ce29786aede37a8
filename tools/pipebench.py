#!/usr/bin/env python3
"""Development aid: frame throughput with N contexts (own streams and workspaces) rendering alternate frames."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--size", type=int, default=1000000)
a = ap.parse_args()
w, h = 1920, 1080
scene = rt.HostScene("soup", w, h, a.size, 1, 0, tight_bvh=True)
ctxs = [rt.Context(0) for _ in range(a.contexts)]
res = []
for c in ctxs:
    ds = c.upload(scene)
    surf, pitch = c.surface(w, h)
    res.append((ds, surf, pitch))
for _ in range(3):
    for c, (ds, surf, pitch) in zip(ctxs, res):
        c.render(ds, scene.opts, scene.camera, surf, pitch)
for c in ctxs:
    c.sync()
t0 = time.perf_counter()
for i in range(a.steps):
    c = ctxs[i % a.contexts]
    ds, surf, pitch = res[i % a.contexts]
    c.render(ds, scene.opts, scene.camera, surf, pitch)
for c in ctxs:
    c.sync()
dt = time.perf_counter() - t0
print(f"contexts {a.contexts}: {dt/a.steps*1e3:.3f} ms per frame -> {3220842/(dt/a.steps)/1e6:.1f} Mrays/s")
for c in ctxs:
    c.close()
