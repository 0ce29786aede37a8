#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow) at 1080p on the 1M-triangle
random soup, on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one full 1920x1080 frame through rt_render (the C-ABI).  With N > 1
the scene blob is broadcast once over RCCL, every rank renders the 8x8 tiles
t = rank, rank+N, ... of the same frame, and the tiles are gathered to rank 0
inside the timed region (strong scaling: the frame is fixed, the work per GPU
shrinks).  Rank 0 prints ONE JSON line.
"""
import os

# Frames in flight run on separate HIP streams, which HIP multiplexes onto this many hardware queues (its default
# is 4, shared with torch's and RCCL's streams).  Must be set before the first HIP call of the process.
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_BENCH_HW_QUEUES", "32")

import argparse  # noqa: E402
import ctypes as C  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cudaraytracer_amd as rt  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
FP32_VECTOR_PEAK_TFLOPS = 256 * 128 * 2 * 2.4e9 / 1e12  # 256 CUs x 128 fp32 lanes x FMA at 2.4 GHz
NODE_BYTES, TRI_BYTES, SPHERE_BYTES, MESH_ENTRY_BYTES = 32, 48, 20, 64
SHADE_MESH_BYTES, SHADE_SPHERE_BYTES, PIXEL_BYTES = 216, 32, 16


def algorithmic_bytes(cnt, pixels):
    """Bytes the traversal must touch: per test / per shaded hit / per pixel figures x the kernel's exact test counts."""
    return (NODE_BYTES * cnt["node"] + TRI_BYTES * cnt["tri"] + SPHERE_BYTES * cnt["sphere"] + MESH_ENTRY_BYTES * cnt["mesh_entry"] +
            SHADE_MESH_BYTES * cnt["shade_mesh"] + SHADE_SPHERE_BYTES * cnt["shade_sphere"] + PIXEL_BYTES * pixels)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scene", default="soup")
    ap.add_argument("--size", type=int, default=1000000, help="triangles (soup / heightfield) or grid side (cubes)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1)
    ap.add_argument("--max-depth", type=int, default=0)
    ap.add_argument("--reference-bvh", action="store_true", help="build BVHs with the reference's FLT_MIN-clamped boxes")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames that may overlap on a GPU (1 = strictly in order, at most 16; default 12)")
    return ap.parse_args()


def cpu_baseline(scene, args):
    """The CPU restatement timed on this host: whole frames of the same workload, repeated until ~cpu_seconds have
    passed, half of them with the build that defines parity (-O2, no contraction), half with the -O3 AVX2 + FMA build
    of the same source where the CPU can run it.  `value` is the faster of the two."""
    from oracle import pyoracle

    threads = max(1, min(16, os.cpu_count() or 1))
    w, h = args.width, args.height
    builds = [("-O2 -ffp-contract=off", False)]
    if pyoracle.o3_supported():
        builds.append(("-O3 -march=x86-64-v3", True))
    figures = []
    for flags, o3 in builds:
        frames, rays, dt = 0, 0, 0.0
        while dt < args.cpu_seconds / len(builds) and frames < 1000:
            t0 = time.perf_counter()
            _, _, cnt = pyoracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=threads, shadow_any_hit=True,
                                        want_keys=False, want_counters=True, o3=o3)
            dt += time.perf_counter() - t0
            rays += cnt["primary"] + cnt["shadow"]
            frames += 1
        figures.append((rays / dt / 1e6, flags, frames, rays, dt))
    best = max(figures)
    return {"value": best[0], "unit": "Mrays/s", "cores": threads, "kind": "port",
            "builds": {f[1]: round(f[0], 3) for f in figures},
            "sample": "; ".join(f"{f[2]} full {w}x{h} frame(s) of the same scene ({f[3]} rays, {f[4]:.1f} s) with gcc {f[1]}" for f in figures) +
                      f"; rows interleaved over {threads} threads, any-hit shadow rays, same BVH as the GPU run; value = the faster build"}


def pmc_table(workload_key):
    """Per-kernel figures of the committed PMC passes for this workload (profiles/traffic.json), or {}."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload_key, {})
    except (OSError, ValueError):
        return {}


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed PMC passes, or None."""
    entry = pmc_table(workload_key).get(kernel)
    return None if entry is None else entry.get("hbm_bytes_per_launch")


def pmc_valu_issue_fraction(workload_key, ms_per_step, cu_count=256, clock_hz=2.4e9):
    """Share of the SIMDs' issue cycles that the frame's vector instructions take: every launch of a frame (SQ_INSTS_VALU
    of the committed PMC pass, 4 cycles per wavefront-wide instruction) over 4 SIMDs x CUs x clock x frame time."""
    table = pmc_table(workload_key)
    frame_kernels = ["traceKernel<false, false>", "traceKernel<true, false>", "shadeKernel<false>", "accumKernel", "finalizeKernel"]
    insts = [table.get(k, {}).get("valu_insts_per_launch") for k in frame_kernels]
    if any(v is None for v in insts[:2]) or ms_per_step <= 0:
        return None
    return 4.0 * sum(v for v in insts if v) / (4 * cu_count * clock_hz * ms_per_step * 1e-3)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch

    # RT_BENCH_FORCE_MULTI=1 (development): take the multi-rank code path with a world of one, e.g. to run its RCCL calls on a one-GPU box
    multi = world > 1 or os.environ.get("RT_BENCH_FORCE_MULTI") == "1"
    # Rehearsal mode for a box with ONE GPU: every rank uses device 0 and the ranks talk over gloo through host
    # memory.  It exercises this script's multi-rank control flow; its numbers mean nothing.
    rehearsal = world > 1 and os.environ.get("RT_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    comm_dev = torch.device("cpu") if rehearsal else dev
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w, h = args.width, args.height
    ctx = rt.Context(dev_index)
    # One explicit stream carries everything this process enqueues through torch (copies, RCCL transfers) and is the
    # library's context stream.  (Torch's default stream is HIP's NULL stream, which the C-ABI reads as "the
    # context's own stream" and which does not order itself against the library's non-blocking streams.)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    # ---- scene: built on rank 0 by the C++ host API, broadcast as one blob ----
    scene = None
    t_build = time.perf_counter()
    if rank == 0:
        scene = rt.HostScene(args.scene, w, h, args.size, args.lights, args.max_depth, tight_bvh=not args.reference_bvh)
    t_build = time.perf_counter() - t_build
    if multi:
        from cudaraytracer_amd import multi as mg

        blob0 = opts0 = cam0 = None
        if rank == 0:
            blob0 = torch.from_numpy(ctx.pack(scene.desc)).to(comm_dev)
            opts0 = bytes(scene.opts.contents)
            cam = scene.camera.contents
            cam0 = list(cam.orig) + list(cam.camera)
        blob, opts_raw, cam_f = mg.share_scene(blob0, opts0, cam0, rank, world, comm_dev)  # RCCL over xGMI: scene + BVH, once
        blob = blob.to(dev)
        torch.cuda.synchronize(dev)
        dscene = ctx.adopt_blob(blob.data_ptr(), blob.numel())
        opts = rt.RtDeviceOpts.from_buffer_copy(opts_raw)
        camera = rt.RtCamera()
        for i in range(3):
            camera.orig[i] = float(cam_f[i])
        for i in range(16):
            camera.camera[i] = float(cam_f[3 + i])
        camera_ref = C.byref(camera)
    else:
        dscene = ctx.upload(scene)
        opts = scene.opts.contents
        camera_ref = scene.camera

    # ---- frame buffers (torch owns the memory; the library gets raw pointers) ----
    # One surface per frame in flight, like a swap chain: frame i renders into surface i % F while
    # frame i-1 is still draining and (with N > 1) being gathered.
    nf = max(1, min(args.frames_in_flight, 16)) if args.frames_in_flight > 0 else 12
    surfaces = [torch.zeros((h, w, 4), dtype=torch.float32, device=dev) for _ in range(nf)]
    surface = surfaces[0]
    pitch = w * 16
    gatherer = mg.TileGather(w, h, rank, world, comm_dev) if multi else None
    issued = [0]

    def collect(frames_ago):
        """Orders torch's stream behind the frame issued `frames_ago` calls ago and gathers its tiles to rank 0
        (where the frame then sits as 8x8-pixel tiles, rank by rank; TileGather.assembled() lays it out as rows)."""
        ctx.stream_wait_frame(stream.cuda_stream, frames_ago)
        gatherer.gather(surfaces[(issued[0] - 1 - frames_ago) % nf].to(comm_dev), assemble=False)

    def step():
        ctx.render(dscene, opts, camera_ref, surfaces[issued[0] % nf].data_ptr(), pitch, None, None, rank, world)
        issued[0] += 1
        if multi and issued[0] >= nf:
            collect(nf - 1)  # the oldest frame in flight; the newer ones keep running meanwhile

    def drain():
        if multi:
            for ago in range(min(nf - 1, issued[0]) - 1, -1, -1):
                collect(ago)
        issued[0] = 0

    # ---- exact ray / test counts of the frame (untimed counter build) ----
    cnt_buf = torch.zeros(rt.RT_COUNTER_CNT, dtype=torch.int64, device=dev)
    ctx.render(dscene, opts, camera_ref, surface.data_ptr(), pitch, None, cnt_buf.data_ptr(), rank, world)
    torch.cuda.synchronize(dev)
    if multi:
        cnt_buf = cnt_buf.to(comm_dev)
        dist.all_reduce(cnt_buf)
    cnt = {name: int(cnt_buf[i].item()) for i, name in enumerate(rt.RT_COUNTER_NAMES)}
    rays = cnt["primary"] + cnt["shadow"]

    ctx.sync()
    ctx.set_frames_in_flight(nf)
    # set-up, not warm-up: one frame per slot, so that every frame in flight has its stream and workspace (a few ms of
    # allocations each) before anything is timed, whatever --warmup says
    for _ in range(nf):
        step()
    drain()
    ctx.sync()
    for _ in range(args.warmup):
        step()
    drain()
    ctx.sync()
    torch.cuda.synchronize(dev)
    ctx.timing_reset()
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    issue_s = time.perf_counter() - t0  # host time to enqueue the frames (and their gathers); the GPU may lag behind
    drain()
    ctx.sync()
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms, stages, launches = ctx.timing_read_stages()
    launches = max(launches, 1)
    # per-stage averages measured with HIP events on the library's stream: [closest-hit trace, shade, shadow trace, rest]
    t_stage = [kernel_ms / launches] + [x / launches for x in stages]
    if multi:
        t = torch.tensor([elapsed, issue_s] + t_stage, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, issue_s, t_stage = float(t[0].item()), float(t[1].item()), [float(x) for x in t[2:].tolist()]
    frame_ms, closest_ms, shade_ms, shadow_ms, rest_ms = t_stage

    # ---- the kernels on their own (untimed for `value`): with frames in flight a launch shares the GPU with the
    # neighbouring frames' kernels and its start-to-end time says little about the kernel itself, so the roofline
    # uses launches that run alone, one frame at a time - the state a rocprofv3 kernel trace of
    # `bench.py --frames-in-flight 1` records.
    alone = t_stage
    if nf > 1:
        ctx.set_frames_in_flight(1)
        for _ in range(2):
            ctx.render(dscene, opts, camera_ref, surfaces[0].data_ptr(), pitch, None, None, rank, world)
        ctx.sync()
        ctx.timing_reset()
        for _ in range(10):
            ctx.render(dscene, opts, camera_ref, surfaces[0].data_ptr(), pitch, None, None, rank, world)
        ctx.sync()
        k_ms, st, n = ctx.timing_read_stages()
        alone = [k_ms / max(n, 1)] + [x / max(n, 1) for x in st]
        if multi:
            t = torch.tensor(alone, dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            alone = [float(x) for x in t.tolist()]
    # ---- the gathered frame against the same frame rendered whole on rank 0 (untimed) ----
    gather_check = None
    if multi and rank == 0:
        whole = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        ctx.render(dscene, opts, camera_ref, whole.data_ptr(), pitch, None, None, 0, 1)
        ctx.sync()
        torch.cuda.synchronize(dev)
        a, b = gatherer.assembled().to(dev), whole
        differing = int(((a != b) & ~(torch.isnan(a) & torch.isnan(b))).any(dim=2).sum().item())
        gather_check = "identical to the frame rendered whole on GPU 0" if differing == 0 else f"{differing} pixels differ from the frame rendered whole on GPU 0"
    in_flight_ms = {"frame": frame_ms, "trace_closest": closest_ms, "shade": shade_ms, "trace_shadow": shadow_ms, "rest": rest_ms}
    frame_ms, closest_ms, shade_ms, shadow_ms, rest_ms = alone

    if rank == 0:
        raw = [int(x) for x in cnt_buf.tolist()]
        sh = dict(zip(("node", "tri", "sphere", "mesh_entry"), raw[9:13]))
        closest = {k: cnt[k] - sh[k] for k in sh}
        # algorithmic bytes of the dominant launch: the closest-hit trace of the primary rays (one launch per frame
        # when maxDepth is 0) reads 32 B per box test, 48 B per triangle test, 20 B per sphere test, 64 B per mesh
        # entry, and writes 20 B of hit record per ray; with N GPUs one launch handles 1/N of the frame
        closest_bytes = (NODE_BYTES * closest["node"] + TRI_BYTES * closest["tri"] + SPHERE_BYTES * closest["sphere"] +
                         MESH_ENTRY_BYTES * closest["mesh_entry"] + 20 * (cnt["primary"] + cnt["secondary"]))
        shadow_bytes = (NODE_BYTES * sh["node"] + TRI_BYTES * sh["tri"] + SPHERE_BYTES * sh["sphere"] + MESH_ENTRY_BYTES * sh["mesh_entry"] +
                        (64 + 1) * cnt["shadow"])
        frame_bytes = algorithmic_bytes(cnt, w * h)
        gbs = lambda b, ms: b / world / (ms * 1e-3) / 1e9 if ms > 0 else None
        achieved = gbs(closest_bytes, closest_ms)
        wl_key = f"{args.scene}-{args.size}-{w}x{h}-l{args.lights}-d{args.max_depth}"
        out = {
            "metric": "Mrays/s (primary+shadow) at 1080p, 1M tris",
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "host_issue_ms_per_step": issue_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene}: {args.size} random triangles in one mesh BVH, {w}x{h}, {args.lights} light(s), "
                                   f"primary + shadow rays, samples 1, maxDepth {args.max_depth}, "
                                   f"{'reference (FLT_MIN-clamped)' if args.reference_bvh else 'tight-bounds'} median-split BVH",
                       "rays_per_frame": rays, "primary": cnt["primary"], "shadow": cnt["shadow"],
                       "tile_sharding": f"8x8 tiles, tile t on GPU t % {world}, gathered to GPU0 each frame" if multi else "single GPU",
                       "frames_in_flight": nf,
                       "gathered_frame": gather_check,
                       "scene_build_s": round(t_build, 2)},
            "kernel_ms": {"frame": frame_ms, "trace_closest": closest_ms, "shade": shade_ms, "trace_shadow": shadow_ms, "rest": rest_ms,
                          "note": "launches running alone (one frame in flight), measured after the timed region"},
            "kernel_ms_in_flight": in_flight_ms,
            "roofline": {"bound": "hbm", "kernel": "traceKernel<false,false> (closest hit, primary rays)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("traceKernel<false, false>", wl_key) if world == 1 else None,
                         "algorithmic_bytes_per_launch": closest_bytes // world,
                         "other_kernels": {"traceKernel<true,false> (shadow, any hit)": {"achieved": gbs(shadow_bytes, shadow_ms),
                                                                                         "algorithmic_bytes_per_launch": shadow_bytes // world},
                                           "whole frame, by wall clock": {"achieved": gbs(frame_bytes, elapsed / args.steps * 1e3),
                                                                          "algorithmic_bytes": frame_bytes // world}},
                         "note": "algorithmic bytes = 32/box test + 48/triangle test + 20/sphere test + 64/mesh entry + 20/ray hit record, "
                                 "test counts exact from the counter build of the same kernels; duration = average of 10 launches running "
                                 "alone (one frame in flight), HIP events on the launch stream; kernel_ms_in_flight holds the start-to-end "
                                 "times inside the timed, overlapped region. "
                                 "The BVH and triangle records (about 56 MB) stay L2 / Infinity-Cache resident, so HBM is the nominal bound only: "
                                 "the figure counts the operands of every test, the caches serve most of them (frac can exceed 1), and "
                                 "`traffic` is what the memory side really moved per launch (PMC)",
                         "tests_per_ray": {"node": cnt["node"] / rays, "tri": cnt["tri"] / rays,
                                           "closest_node": closest["node"] / max(cnt["primary"] + cnt["secondary"], 1),
                                           "closest_tri": closest["tri"] / max(cnt["primary"] + cnt["secondary"], 1),
                                           "shadow_node": sh["node"] / max(cnt["shadow"], 1), "shadow_tri": sh["tri"] / max(cnt["shadow"], 1)}},
            # second bound of this path: fp32 vector work.  24 flops per box test, 60 per triangle test, 30 per sphere
            # test, 31 per mesh entry, over the frame's wall-clock time; peak = CUs x 128 lanes x 2 (FMA) x clock
            "valu": {"achieved": (24 * cnt["node"] + 60 * cnt["tri"] + 30 * cnt["sphere"] + 31 * cnt["mesh_entry"]) / world /
                                 (elapsed / args.steps) / 1e12,
                     "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     # measured counterpart: SIMD issue cycles taken by the frame's vector instructions (PMC pass of the same
                     # workload) over the cycles available in this run's frame time; None without that pass or with N > 1
                     "issue_fraction": pmc_valu_issue_fraction(wl_key, elapsed / args.steps * 1e3) if world == 1 else None,
                     "note": "achieved = algorithmic flops of the tests (no selects, compares, address or stack work; no FMA "
                             "contraction, so half the peak is out of reach by construction); issue_fraction says how busy the "
                             "vector units really are"},
        }
        if not args.no_cpu and args.cpu_seconds > 0 and not multi:
            out["cpu_baseline"] = cpu_baseline(scene, args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
