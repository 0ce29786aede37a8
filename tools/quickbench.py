#!/usr/bin/env python3
"""Stage timing of the frame pipeline on one GPU (development aid).

  python tools/quickbench.py [--scene soup] [--size 1000000] [--lights 1] [--steps 10]
"""
import argparse
import os
import sys
import time

os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_BENCH_HW_QUEUES", "32")  # frames in flight need more hardware queues than the 4 a box exports

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="soup")
ap.add_argument("--size", type=int, default=1000000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lights", type=int, default=1)
ap.add_argument("--max-depth", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--reference-bvh", action="store_true")
ap.add_argument("--lds-stack", type=int, default=0)
ap.add_argument("--frames-in-flight", type=int, default=0)
ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--skip-unlit", action="store_true", help="rt_set_unlit_shadow_skip: shadow rays towards lights behind the surface are not walked")
ap.add_argument("--split", type=int, default=-1, help="rt_set_box_test_split: level-0 traces of at least this many rays run as two launches")
ap.add_argument("--cost-order", action="store_true", help="tiles in order of falling cost, measured by a counter frame (rt_set_tile_list)")
ap.add_argument("--tile-order", type=int, default=-1)
ap.add_argument("--cost-first", type=float, default=0.0,
                help="with --cost-order: only this fraction of the tiles (the most expensive) goes first, in order of falling cost; the rest keeps the centre-out order")
ap.add_argument("--device-bvh", action="store_true")
ap.add_argument("--device-sah", action="store_true", help="with --device-bvh: the device cuts by the binned surface-area heuristic (rt_set_device_bvh_rule), leaves of 8")
ap.add_argument("--sah", action="store_true", help="mesh BVHs cut by the binned surface-area heuristic (host build)")
ap.add_argument("--leaf", type=int, default=0, help="most triangles in a leaf (0: builder default)")
ap.add_argument("--assets", default=None, help="asset directory for the mixed / barrel scenes; 'synthetic' writes stand-ins to a temp dir")
ap.add_argument("--stride", type=int, default=1, help="render only tiles 0, stride, 2*stride, ... (one GPU's share of a multi-GPU frame)")
a = ap.parse_args()

assets = a.assets
if assets == "synthetic":
    import tempfile

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import synth_assets

    assets = synth_assets.write_assets(tempfile.mkdtemp(prefix="rt_assets_"))
t = time.time()
scene = rt.HostScene(a.scene, a.width, a.height, a.size, a.lights, a.max_depth, asset_dir=assets, tight_bvh=not a.reference_bvh,
                     device_bvh=a.device_bvh, sah_bvh=a.sah, leaf_threshold=a.leaf)
print(f"scene {a.scene} size {a.size}: built in {time.time()-t:.2f}s, depth {scene.bvh_depth()}", flush=True)
with rt.Context(0) as ctx:
    if a.waves:
        ctx.set_trace_waves(a.waves)
    if a.split >= 0:
        ctx.set_box_test_split(a.split)
    if a.skip_unlit:
        ctx.set_unlit_shadow_skip(True)
    if a.tile_order >= 0:
        ctx.set_tile_order(a.tile_order)
    if a.frames_in_flight:
        ctx.set_frames_in_flight(a.frames_in_flight)
    if a.lds_stack:
        ctx.set_lds_stack(a.lds_stack)
    ctx.set_device_bvh(a.leaf or (8 if a.device_sah else 10), not a.reference_bvh)
    ctx.set_device_bvh_rule(a.device_sah)
    ctx.sync()
    t = time.time()
    ds = ctx.upload(scene)
    ctx.sync()
    print(f"upload{' + device BVH build' if a.device_bvh else ''}: {time.time()-t:.3f}s", flush=True)
    t = time.time()
    ds2 = ctx.upload(scene)
    ctx.sync()
    print(f"second upload: {time.time()-t:.3f}s", flush=True)
    ctx.free_scene(ds2)
    surf, pitch = ctx.surface(a.width, a.height)
    cnt = ctx.alloc(rt.RT_COUNTER_CNT * 8 * 8)  # room for the diagnostic build's section stamps
    ctx.render(ds, scene.opts, scene.camera, surf, pitch, None, cnt)
    ctx.sync()
    import numpy as np
    raw = ctx.download(cnt, (rt.RT_COUNTER_CNT,), np.uint64)
    c = {n: int(raw[i]) for i, n in enumerate(rt.RT_COUNTER_NAMES)}
    for name, b in (("closest", 16), ("shadow", 22)):
        st = [int(v) for v in raw[b:b + 6]]
        if st[0]:
            tl = [int(v) for v in raw[28 + (0 if name == "closest" else 2):30 + (0 if name == "closest" else 2)]]
            print(f"  {name}: tail {100.0*tl[0]/st[0]:.1f}% of the wave-steps with {tl[1]/max(tl[0],1):.1f} lanes; before it {(st[5]-tl[1])/max(st[0]-tl[0],1):.1f} lanes")
            print(f"  {name}: wave-steps {st[0]}  node-steps {st[1]} (avg lanes {st[2]/max(st[1],1):.1f})  tri-steps {st[3]} (avg lanes {st[4]/max(st[3],1):.1f})  avg lanes with a ray {st[5]/st[0]:.1f}")
    if os.environ.get("RT_STAMPS"):  # a library built with -DRT_STAMPS: cycles per loop section, summed over wavefronts
        big = ctx.download(cnt, (128,), np.uint64)
        names = ["refill", "vote", "node: wait for pair", "node: math + push", "tri: publish", "tri: wait for records", "tri: math + store",
                 "tri: apply", "shapes", "pops", "publish"]
        for kname, base in (("closest", 64), ("shadow", 96)):
            tot = sum(int(big[base + 2 * i]) for i in range(16))
            print(f"  {kname}: {tot/1e6:.1f} M cycles in all wavefronts")
            for i, n in enumerate(names):
                cyc, k = int(big[base + 2 * i]), int(big[base + 2 * i + 1])
                print(f"    {n:24s} {100.0*cyc/max(tot,1):5.1f} %  {k:9d} times  {cyc/max(k,1):8.1f} cycles each")
    if a.cost_order:
        tx, ty = (a.width + 7) // 8, (a.height + 7) // 8
        cost = ctx.alloc(tx * ty * 4)
        ctx.set_tile_cost_buffer(cost)
        ctx.render(ds, scene.opts, scene.camera, surf, pitch, None, cnt)
        ctx.sync()
        ctx.set_tile_cost_buffer(None)
        costs = ctx.download(cost, (tx * ty,), np.uint32).astype(np.int64)
        order = np.argsort(-costs, kind="stable")[:: 1]
        if a.cost_first > 0.0:
            head = order[: max(1, int(len(order) * a.cost_first))]
            ys, xs = np.divmod(np.arange(tx * ty), tx)
            d2 = (xs * 16 + 8 - a.width) ** 2 + (ys * 16 + 8 - a.height) ** 2  # the library's centre-out key
            centre_out = np.lexsort((np.arange(tx * ty), d2))
            taken = np.zeros(tx * ty, dtype=bool)
            taken[head] = True
            order = np.concatenate([head, centre_out[~taken[centre_out]]])
        order = order[a.stride - 1:: a.stride] if a.stride > 1 else order
        ctx.set_tile_list(order)
        print(f"  cost order: most expensive tile {costs.max()} tests, median {int(np.median(costs))}, cheapest {costs.min()}")
    rays = c["primary"] + c["shadow"] + c["secondary"]
    nf = max(a.frames_in_flight, 1)
    surfs = [surf] + [ctx.surface(a.width, a.height)[0] for _ in range(nf - 1)]
    for i in range(2 * nf):
        ctx.render(ds, scene.opts, scene.camera, surfs[i % nf], pitch, None, None, 0, 1 if a.cost_order else a.stride)
    ctx.sync()
    ctx.timing_reset()
    t0 = time.perf_counter()
    issue = 0.0
    for i in range(a.steps):
        ti = time.perf_counter()
        ctx.render(ds, scene.opts, scene.camera, surfs[i % nf], pitch, None, None, 0, 1 if a.cost_order else a.stride)
        issue += time.perf_counter() - ti
    print(f"host time inside rt_render: {issue/a.steps*1e3:.3f} ms per frame")
    ctx.sync()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    try:
        total, st, n = ctx.timing_read_stages()
    except rt.RtError:  # no stage events (an experiment path that does not record them)
        total, st, n = 0.0, [0.0] * 4, 1
    n = max(n, 1)
    print(f"rays/frame {rays} (primary {c['primary']} shadow {c['shadow']} secondary {c['secondary']}), node/ray {c['node']/rays:.1f} tri/ray {c['tri']/rays:.1f}")
    print(f"frame {total/n:.3f} ms  | primary-trace {st[0]/n:.3f}  shade {st[1]/n:.3f}  shadow-trace {st[2]/n:.3f}  rest {st[3]/n:.3f}  | "
          f"wall {wall:.3f} ms/frame = {rays/a.stride/wall/1e3:.1f} Mrays/s", flush=True)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.download_surface(surfs[0], pitch, a.width, a.height)
    print(f"surface download (pageable host memory): {(time.perf_counter()-t0)/5*1e3:.3f} ms per frame")
