#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow) at 1080p on the 1M-triangle
random soup, on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one full 1920x1080 frame through rt_render (the C-ABI).  With N > 1
the scene blob is shared once over RCCL, every rank renders the 8x8 tiles
t = rank, rank+N, ... of the same frame, and the tiles are gathered to rank 0
inside the timed region (strong scaling: the frame is fixed, the work per GPU
shrinks).  Rank 0 prints ONE JSON line.

`value` is measured with frames in flight; the same line carries the strictly
in-order figure, the figure a process with HIP's default four hardware queues
gets, the other workloads of the scope table, the roofline of the dominant
kernel and the CPU baseline.
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
FP32_VECTOR_PEAK_TFLOPS = 256 * 4 * 32 * 2 * 2.4e9 / 1e12  # 256 CUs x 4 SIMDs x 32 fp32 lanes x 2 (FMA) at 2.4 GHz = 157.3
# algorithmic work per test (the survey's per-unit figures)
NODE_FLOPS, TRI_FLOPS, SPHERE_FLOPS, MESH_ENTRY_FLOPS = 24, 60, 30, 31
HIT_RECORD_BYTES = 20  # float4 {t, u, v, shape} + int triangle, written per traced path node
PROFILE_ROUND = "r03"


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
    ap.add_argument("--bvh", choices=("sah", "median"), default="sah",
                    help="mesh BVH of the headline scene: cut by the binned surface-area heuristic (leaves <= 8 triangles) or by the "
                         "reference's median cut (leaves <= 10); same image either way")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--tile-balance", choices=("cost", "interleaved"), default="cost",
                    help="N > 1: tiles per GPU by measured cost (equal box + triangle tests per GPU) or tile t on GPU t %% N")
    ap.add_argument("--no-others", action="store_true", help="skip the other workloads and the default-hardware-queues run (headline only)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames that may overlap on a GPU (1 = strictly in order, at most 16; default 12)")
    return ap.parse_args()


def host_threads():
    """Threads for the CPU baseline: every CPU this process may run on, within the share the box grants a job
    (OMP_NUM_THREADS is set to that share on a shared machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    grant = os.environ.get("OMP_NUM_THREADS", "")
    if grant.isdigit() and int(grant) > 0:
        n = min(n, int(grant))
    return max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, args):
    """The CPU restatement timed on this host: whole frames of the same workload, repeated until ~cpu_seconds have
    passed, half of them with the build that defines parity (-O2, no contraction), half with the -O3 AVX2 + FMA build
    of the same source where the CPU can run it.  `value` is the faster of the two."""
    from oracle import pyoracle

    threads = host_threads()
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
            "host_logical_cpus": os.cpu_count(), "cpu_model": cpu_model(),
            "builds": {f[1]: round(f[0], 3) for f in figures},
            "sample": "; ".join(f"{f[2]} full {w}x{h} frame(s) of the same scene ({f[3]} rays, {f[4]:.1f} s) with gcc {f[1]}" for f in figures) +
                      f"; rows interleaved over {threads} threads (every CPU this job is granted; the machine has {os.cpu_count()} "
                      "logical CPUs, shared with other jobs), any-hit shadow rays, same BVH as the GPU run; value = the faster build"}


def profile_table(workload_key):
    """Per-kernel figures of the committed PMC passes for this workload (profiles/traffic.json), or {}.  These are
    constants measured in an earlier, separate rocprofv3 run of this script - not measurements of the run that prints them."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload_key, {})
    except (OSError, ValueError):
        return {}


def counters_of(ctx, dscene, opts, camera_ref, surface_ptr, pitch, rank=0, world=1):
    """One untimed frame with the counter build of the kernels: exact ray / test counts and the wave statistics."""
    import torch

    cnt_buf = torch.zeros(rt.RT_COUNTER_CNT, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
    ctx.render(dscene, opts, camera_ref, surface_ptr, pitch, None, cnt_buf.data_ptr(), rank, world)
    ctx.sync()
    torch.cuda.synchronize()
    return cnt_buf


def wave_statistics(raw):
    """Lane use of the two trace kernels, from the counter build's own statistics (measured in this run)."""
    out = {}
    for name, base, tail in (("trace_closest", 16, 28), ("trace_shadow", 22, 30)):
        steps, node_steps, node_lanes, tri_steps, tri_rays, busy = (int(v) for v in raw[base:base + 6])
        tail_steps, tail_lanes = int(raw[tail]), int(raw[tail + 1])
        if steps == 0:
            continue
        out[name] = {
            "wave_steps": steps,
            "node_steps": node_steps, "lanes_per_node_step": node_lanes / max(node_steps, 1),
            "triangle_phases": tri_steps, "rays_per_triangle_phase_of_16": tri_rays / max(tri_steps, 1),
            "lanes_holding_a_ray": busy / steps,
            "steps_after_queue_ran_dry": tail_steps / steps,
            "lanes_holding_a_ray_after_queue_ran_dry": tail_lanes / max(tail_steps, 1),
        }
    return out


def time_frames(ctx, render, frames, in_flight, surfaces):
    """(wall ms per frame, HIP-event ms per frame [frame, closest, shade, shadow, rest]) of `frames` calls of render(surface)."""
    import torch

    ctx.sync()
    ctx.set_frames_in_flight(in_flight)
    for i in range(max(in_flight, 2)):  # every slot gets its stream and workspace before anything is timed
        render(surfaces[i % len(surfaces)])
    ctx.sync()
    ctx.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        render(surfaces[i % len(surfaces)])
    ctx.sync()
    wall = (time.perf_counter() - t0) / frames * 1e3
    k_ms, st, n = ctx.timing_read_stages()
    n = max(n, 1)
    return wall, [k_ms / n] + [x / n for x in st]


def roofline_factors(prof, flops):
    """The roofline fraction as a product of three measured factors (committed PMC passes of the same workload): how often
    a SIMD issues a vector instruction against the peak's one per two cycles, how many of the 64 lanes an instruction runs
    with, and how many of the counted flops an active lane's instruction carries against the peak's two (a fused multiply-add)."""
    need = ("valu_insts_per_launch", "valu_thread_cycles_per_launch", "gui_active_cycles_per_launch")
    if not prof or any(k not in prof for k in need):
        return None
    insts, lanes, cycles = (prof[k] for k in need)
    simd_cycles = 256 * 4 * cycles / 8  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    issue, lane_use, carried = insts / simd_cycles / 0.5, lanes / insts / 64, flops / lanes / 2
    return {"issue_rate_vs_peak": issue, "lanes_per_instruction_of_64": lane_use, "flops_per_active_lane_instruction_of_2": carried,
            "product": issue * lane_use * carried,
            "note": "product = achieved / peak at the profiled launch's clock (GRBM_GUI_ACTIVE / 8 XCDs over its duration); counters from the "
                    "committed rocprofv3 passes, flops from this run's counter build"}


def other_workload(ctx, torch, dev, label, scene_args, scene_kwargs, nf):
    """In-order and in-flight figures of another workload of the scope table, on the same kernels.
    scene_kwargs["upload_refit"]: upload with rt_set_upload_refit on (boxes recomputed on the device);
    scene_kwargs["upload_rebuild"]: with rt_set_upload_rebuild on (mesh BVHs built again on the device);
    scene_kwargs["unlit_skip"]: timed with rt_set_unlit_shadow_skip on - the row then states frame times only, because
    fewer rays are walked than the counters (which follow the reference) say."""
    scene_kwargs = dict(scene_kwargs)
    refit = scene_kwargs.pop("upload_refit", False)
    rebuild = scene_kwargs.pop("upload_rebuild", False)
    unlit_skip = scene_kwargs.pop("unlit_skip", False)
    device_sah = scene_kwargs.pop("device_sah", 0)  # leaf threshold: the device cuts the meshes it builds by the binned surface-area heuristic
    t0 = time.perf_counter()
    scene = rt.HostScene(*scene_args, **scene_kwargs)
    build_s = time.perf_counter() - t0
    w, h = scene.width, scene.height
    ctx.set_upload_refit(refit)
    ctx.set_upload_rebuild(rebuild)
    if device_sah:
        ctx.set_device_bvh(device_sah, True)
        ctx.set_device_bvh_rule(sah=True)
    t0 = time.perf_counter()
    try:
        dscene = ctx.upload(scene)
        ctx.sync()
    finally:
        ctx.set_upload_refit(False)
        ctx.set_upload_rebuild(False)
        ctx.set_device_bvh_rule(sah=False)
        ctx.set_device_bvh(10, True)
    upload_s = time.perf_counter() - t0
    surfaces = [torch.zeros((h, w, 4), dtype=torch.float32, device=dev) for _ in range(nf)]
    pitch = w * 16
    opts, cam = scene.opts.contents, scene.camera
    raw = counters_of(ctx, dscene, opts, cam, surfaces[0].data_ptr(), pitch).tolist()
    cnt = dict(zip(rt.RT_COUNTER_NAMES, raw))
    rays = cnt["primary"] + cnt["shadow"] + cnt["secondary"]

    def render(s):
        ctx.render(dscene, opts, cam, s.data_ptr(), pitch)

    ctx.set_unlit_shadow_skip(unlit_skip)
    try:
        wall1, ev1 = time_frames(ctx, render, 10, 1, surfaces)
        walln, _ = time_frames(ctx, render, 6 * nf, nf, surfaces)  # long enough that filling and draining the slots does not show
    finally:
        ctx.set_unlit_shadow_skip(False)
    ctx.set_frames_in_flight(1)
    ctx.free_scene(dscene)
    scene.close()
    del surfaces
    torch.cuda.empty_cache()
    row = {"workload": label, "rays_per_frame": rays, "primary": cnt["primary"], "shadow": cnt["shadow"], "secondary": cnt["secondary"],
           "box_tests_per_ray": cnt["node"] / max(rays, 1), "triangle_tests_per_ray": cnt["tri"] / max(rays, 1),
           "in_order_ms_per_frame": ev1[0], "in_order_Mrays_per_s": rays / wall1 / 1e3,
           "in_flight_ms_per_frame": walln, "in_flight_Mrays_per_s": rays / walln / 1e3, "frames_in_flight": nf,
           "scene_build_s": round(build_s, 2), "upload_s": round(upload_s, 2)}
    if unlit_skip:
        row["in_order_Mrays_per_s"] = row["in_flight_Mrays_per_s"] = None
        row["note"] = ("rays_per_frame and the tests per ray are the reference's (counter frames walk every ray); the timed frames leave out the shadow "
                       "rays of lights behind the surface, so only the frame times compare with the row without the switch")
    return row


def default_queue_run(args, nf):
    """The headline's timed region again in a child process that leaves GPU_MAX_HW_QUEUES at HIP's default of 4 (the
    variable is read once, at the process's first HIP call, so one process cannot measure both): what a caller gets
    who asks for frames in flight and sets nothing else."""
    import subprocess

    env = dict(os.environ)
    env["RT_BENCH_HW_QUEUES"] = "4"
    steps = max(args.steps // 2, 20)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(args.warmup), "--no-others", "--no-cpu",
           "--scene", args.scene, "--size", str(args.size), "--width", str(args.width), "--height", str(args.height), "--lights", str(args.lights),
           "--max-depth", str(args.max_depth), "--bvh", args.bvh, "--frames-in-flight", str(nf)] + (["--reference-bvh"] if args.reference_bvh else [])
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        child = json.loads(line)
        return {"GPU_MAX_HW_QUEUES": 4, "value": child["value"], "unit": child["unit"], "ms_per_step": child["ms_per_step"], "steps": steps,
                "frames_in_flight": nf, "in_order_Mrays_per_s": child["config"]["in_order"]["Mrays_per_s"],
                "note": "same workload and timed region in a child process with HIP's default number of hardware queues; `value` above ran with "
                        f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}"}
    except Exception as e:  # reported, the headline stays
        return {"GPU_MAX_HW_QUEUES": 4, "error": f"{type(e).__name__}: {e}"}


class mg_single:
    """The swap chain without a gather (one GPU): same step / drain calls as cudaraytracer_amd.multi.FramePipeline."""

    def __init__(self, surfaces, render):
        self.surfaces, self.render, self.issued = surfaces, render, 0

    def step(self):
        self.render(self.surfaces[self.issued % len(self.surfaces)])
        self.issued += 1

    def drain(self):
        self.issued = 0


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
        os.environ.setdefault("MASTER_PORT", "29517")
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

    # ---- scene: built on rank 0 by the C++ host API, shared as one blob ----
    scene = None
    t_build = time.perf_counter()
    if rank == 0:
        scene = rt.HostScene(args.scene, w, h, args.size, args.lights, args.max_depth, tight_bvh=not args.reference_bvh,
                             sah_bvh=args.bvh == "sah" and not args.reference_bvh)
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
        blob = blob.to(dev).contiguous()
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
    # RT_BENCH_TORCH_GATHER=1 (development): pack / assemble tiles with torch index arithmetic instead of the library's kernels
    use_kernels = os.environ.get("RT_BENCH_TORCH_GATHER") != "1"
    gatherer = mg.TileGather(w, h, rank, world, comm_dev, ctx=ctx if use_kernels else None) if multi else None
    if multi:
        # frame i renders into surface i % F; the oldest frame in flight is gathered to rank 0 (where it then sits as
        # 8x8-pixel tiles, rank by rank; TileGather.assembled() lays it out as rows) while the newer ones keep rendering
        pipe = mg.FramePipeline(surfaces, lambda s, i: ctx.render(dscene, opts, camera_ref, s.data_ptr(), pitch, None, None, *shard),
                                lambda ago: ctx.stream_wait_frame(stream.cuda_stream, ago), gatherer,
                                None if use_kernels else (lambda s: s.to(comm_dev)))
    else:
        pipe = mg_single(surfaces, lambda s: ctx.render(dscene, opts, camera_ref, s.data_ptr(), pitch))
    step, drain = pipe.step, pipe.drain

    # ---- exact ray / test counts of the frame (untimed counter build) ----
    shard = [rank, world]  # tileFirst, tileStride of every render call; [0, 1] once a tile list is set
    tile_cost = None
    if multi and args.tile_balance == "cost":
        tile_cost = torch.zeros(gatherer.n, dtype=torch.int32, device=dev)
        ctx.set_tile_cost_buffer(tile_cost.data_ptr())
    cnt_buf = counters_of(ctx, dscene, opts, camera_ref, surface.data_ptr(), pitch, rank, world)
    if multi:
        cnt_buf = cnt_buf.to(comm_dev)
        dist.all_reduce(cnt_buf)
    balance = None
    if tile_cost is not None:
        # Every rank measured its interleaved share of the tiles; summed, that is the frame's cost map.  Every rank
        # derives the same lists from it: equal tile counts, equal box + triangle tests, expensive tiles first.
        ctx.set_tile_cost_buffer(None)
        cost_all = tile_cost.to(comm_dev).to(torch.int64)
        dist.all_reduce(cost_all)
        cost_all = cost_all.cpu()
        lists = [mg.center_out(l, w, h) for l in mg.balanced_assignment(cost_all, world)]  # equal cost per GPU, rendered centre first
        loads = [int(cost_all[l].sum()) for l in lists]
        inter = [int(cost_all[l].sum()) for l in mg.interleaved_assignment(w, h, world)]
        balance = {"max_over_mean_cost_balanced": max(loads) / (sum(loads) / world), "max_over_mean_cost_interleaved": max(inter) / (sum(inter) / world)}
        ctx.set_tile_list(lists[rank].numpy())
        gatherer.set_assignment(lists)
        shard[:] = [0, 1]
    cnt = {name: int(cnt_buf[i].item()) for i, name in enumerate(rt.RT_COUNTER_NAMES)}
    rays = cnt["primary"] + cnt["shadow"]

    def timed_region(steps, warmup):
        """The contract's timed region: `warmup` untimed steps, then exactly `steps` steps between barriers."""
        ctx.sync()
        ctx.set_frames_in_flight(nf)
        # set-up, not warm-up: one frame per slot, so that every frame in flight has its stream and workspace (a few ms of
        # allocations each) before anything is timed, whatever --warmup says
        for _ in range(nf):
            step()
        drain()
        ctx.sync()
        for _ in range(warmup):
            step()
        drain()
        ctx.sync()
        torch.cuda.synchronize(dev)
        ctx.timing_reset()
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
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
        # per-stage averages measured with HIP events on the library's stream: [frame, closest-hit trace, shade, shadow trace, rest]
        t_stage = [kernel_ms / launches] + [x / launches for x in stages]
        if multi:
            t = torch.tensor([elapsed, issue_s] + t_stage, dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, issue_s, t_stage = float(t[0].item()), float(t[1].item()), [float(x) for x in t[2:].tolist()]
        return elapsed, issue_s, t_stage

    def alone(frames=10):
        """The kernels on their own: one frame in flight, strictly in order - the state a rocprofv3 kernel trace of
        `bench.py --frames-in-flight 1` records.  Returns per-frame HIP-event averages [frame, closest, shade, shadow, rest]
        and the wall time per frame."""
        ctx.sync()
        ctx.set_frames_in_flight(1)
        for _ in range(2):
            ctx.render(dscene, opts, camera_ref, surfaces[0].data_ptr(), pitch, None, None, *shard)
        ctx.sync()
        ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(frames):
            ctx.render(dscene, opts, camera_ref, surfaces[0].data_ptr(), pitch, None, None, *shard)
        ctx.sync()
        wall = (time.perf_counter() - t0) / frames * 1e3
        k_ms, st, n = ctx.timing_read_stages()
        out = [k_ms / max(n, 1)] + [x / max(n, 1) for x in st] + [wall]
        if multi:
            t = torch.tensor(out, dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out = [float(x) for x in t.tolist()]
        return out[:5], out[5]

    # ================= the timed region =================
    elapsed, issue_s, in_flight_stage = timed_region(args.steps, args.warmup)
    alone_stage, alone_wall = alone() if nf > 1 else (in_flight_stage, elapsed / args.steps * 1e3)

    # ---- throughput against latency: the same frames with fewer of them in flight (one GPU, after the timed region) ----
    sweep = None
    if not multi and not args.no_others:
        sweep = []
        for k in (2, 3, 4, 6):
            if k >= nf:
                continue
            wall_k, ev_k = time_frames(ctx, lambda s_: ctx.render(dscene, opts, camera_ref, s_.data_ptr(), pitch), 60, k, surfaces)
            sweep.append({"frames_in_flight": k, "Mrays_per_s": rays / wall_k / 1e3, "ms_per_frame": wall_k, "frame_latency_ms": ev_k[0]})
        ctx.set_frames_in_flight(nf)

    # ---- the gathered frame against the same frame rendered whole on rank 0 (untimed) ----
    gather_check = None
    if multi and rank == 0:
        whole = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        if tile_cost is not None:
            ctx.set_tile_list(None)
        ctx.render(dscene, opts, camera_ref, whole.data_ptr(), pitch, None, None, 0, 1)
        ctx.sync()
        if tile_cost is not None:
            ctx.set_tile_list(lists[0].numpy())
        torch.cuda.synchronize(dev)
        a, b = gatherer.assembled().to(dev), whole
        differing = int(((a != b) & ~(torch.isnan(a) & torch.isnan(b))).any(dim=2).sum().item())
        gather_check = "identical to the frame rendered whole on GPU 0" if differing == 0 else f"{differing} pixels differ from the frame rendered whole on GPU 0"

    # ================= other workloads of the scope table (untimed for `value`, one GPU) =================
    others = []
    if not args.no_others and not multi:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import tempfile

        import synth_assets

        ctx.free_scene(dscene)
        dscene = None
        assets = synth_assets.write_assets(tempfile.mkdtemp(prefix="rt_assets_"))
        plans = [
            ("cfg 2 on the reference's median-cut tree (cudaTracer/BVH.h:104-181, leaves <= 10) with tight boxes: the same soup, the same "
             "image, the tree shape the reference builds",
             ("soup", w, h, args.size, args.lights, args.max_depth), {"tight_bvh": True}),
            ("cfg 2 on the reference's own BVH: the same soup with FLT_MIN-clamped boxes (cudaTracer/BoundingBox.cuh:25-27)",
             ("soup", w, h, args.size, args.lights, args.max_depth), {"tight_bvh": False}),
            ("cfg 2 on the reference's own BVH with the boxes recomputed at upload (rt_set_upload_refit): the reference's tree shape, leaf order "
             "and ties, true bounds; its cuts were made on FLT_MIN-clamped centroid bounds (cudaTracer/BVH.h:122-128), so it stays behind a tree "
             "built on true bounds",
             ("soup", w, h, args.size, args.lights, args.max_depth), {"tight_bvh": False, "upload_refit": True}),
            ("cfg 2 handed over with the reference's own BVH, mesh BVHs built again on the device at upload (rt_set_upload_rebuild): the "
             "reference's median cut on true bounds, same image; what a host that passes the reference's arrays gets with one switch",
             ("soup", w, h, args.size, args.lights, args.max_depth), {"tight_bvh": False, "upload_rebuild": True}),
            ("cfg 2 handed over with the reference's own BVH, mesh BVHs cut again on the device by the binned surface-area heuristic "
             "(rt_set_upload_rebuild + rt_set_device_bvh_rule, leaves of 8): the headline's tree, built in 0.06 s at upload",
             ("soup", w, h, args.size, args.lights, args.max_depth), {"tight_bvh": False, "upload_rebuild": True, "device_sah": 8}),
            ("cfg 2 with 10 lights (the workload's second run): the headline's soup and surface-area tree, ten shadow rays per hit",
             ("soup", w, h, args.size, 10, args.max_depth), {"tight_bvh": True, "sah_bvh": True}),
            ("cfg 2 with 10 lights and rt_set_unlit_shadow_skip: shadow rays towards lights behind the surface (lambert 0, nothing to add) are "
             "not walked; same pixels",
             ("soup", w, h, args.size, 10, args.max_depth), {"tight_bvh": True, "sah_bvh": True, "unlit_skip": True}),
            ("cfg 1: 16x16 grid of cube meshes (one mesh BVH each) under a scene BVH, 640x480, 4 lights, maxDepth 5",
             ("cubes", 640, 480, 16, 4, 5), {"tight_bvh": True}),
            ("cfg 3: OBJ barrel (488 triangles, three 1024x1024 maps) in three instances, one a mirror, textured plane, mirror and "
             "glass balls, 1920x1080, 10 lights, maxDepth 2 (stand-in asset files of the reference's shape)",
             ("barrel", 1920, 1080, 0, 10, 2), {"asset_dir": assets, "tight_bvh": True}),
            ("cfg 4: height field of 9 999 392 triangles in one mesh BVH cut by the surface-area heuristic (host build), 3840x2160, 1 light",
             ("heightfield", 3840, 2160, 10000000, 1, 0), {"tight_bvh": True, "sah_bvh": True}),
            ("cfg 4 with the same surface-area tree cut on the device (rt_set_device_bvh_rule; upload_s includes the build)",
             ("heightfield", 3840, 2160, 10000000, 1, 0), {"tight_bvh": True, "device_bvh": True, "device_sah": 8}),
            ("cfg 4 with the reference's median-cut tree, built on the device (rt_set_device_bvh)",
             ("heightfield", 3840, 2160, 10000000, 1, 0), {"tight_bvh": True, "device_bvh": True}),
            ("cfg 5: default-scene analogue (sphere field, mirror and glass balls, textured plane, three mesh instances), "
             "1920x1080, 16 lights, maxDepth 4",
             ("mixed", 1920, 1080, 0, 16, 4), {"asset_dir": assets, "tight_bvh": True}),
        ]
        for label, a, kw in plans:
            try:
                others.append(other_workload(ctx, torch, dev, label, a, kw, nf))
            except Exception as e:  # a workload that cannot run is reported, the headline stays
                others.append({"workload": label, "error": str(e)})

    default_queues = default_queue_run(args, nf) if (rank == 0 and not multi and not args.no_others) else None

    frame_ms, closest_ms, shade_ms, shadow_ms, rest_ms = alone_stage
    if rank == 0:
        raw = [int(x) for x in cnt_buf.tolist()]
        sh = dict(zip(("node", "tri", "sphere", "mesh_entry"), raw[9:13]))
        closest = {k: cnt[k] - sh[k] for k in sh}
        traced = cnt["primary"] + cnt["secondary"]
        flops = lambda c: NODE_FLOPS * c["node"] + TRI_FLOPS * c["tri"] + SPHERE_FLOPS * c["sphere"] + MESH_ENTRY_FLOPS * c["mesh_entry"]
        closest_flops, shadow_flops = flops(closest), flops(sh)
        tflops = lambda f, ms: f / world / (ms * 1e-3) / 1e12 if ms > 0 else None
        achieved = tflops(closest_flops, closest_ms)
        # compulsory HBM bytes of the closest-hit launch: the BVH nodes and triangle records it walks, read once, and the
        # hit records it writes; everything else it touches again comes from L2 / Infinity Cache by construction
        d = scene.desc.contents
        bvh_bytes = 32 * d.nodeCnt + 64 * d.shapeCnt + sum(32 * d.meshes[i].nodeCnt + 48 * d.meshes[i].triangleCnt for i in range(d.meshCnt))
        compulsory = bvh_bytes + HIT_RECORD_BYTES * traced // world
        wl_key = f"{args.scene}-{args.size}-{w}x{h}-l{args.lights}-d{args.max_depth}"
        # (the closest-hit kernel of frames without jitter: template arguments ANY, COUNT, LIT, JIT all false)
        prof = profile_table(wl_key).get("traceKernel<false, false, false, false>", {}) if world == 1 else {}
        prof_src = f"profiles/traffic.json <- profiles/{PROFILE_ROUND}: separate rocprofv3 --pmc passes of `bench.py --frames-in-flight 1`, NOT this run"
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+shadow) at 1080p, 1M tris",
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "host_issue_ms_per_step": issue_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene}: {args.size} random triangles in one mesh BVH, {w}x{h}, {args.lights} light(s), "
                                   f"primary + shadow rays, samples 1, maxDepth {args.max_depth}, "
                                   + ("the reference's median-cut BVH with its FLT_MIN-clamped boxes; " if args.reference_bvh else
                                      "tight-bounds BVH cut by the binned surface-area heuristic, leaves <= 8 triangles (rt_mesh_bvh_build's rule; "
                                      "the reference's median-cut tree is under other_workloads); " if args.bvh == "sah" else
                                      "tight-bounds BVH, the reference's median cut, leaves <= 10; ") +
                                   "fp32 arithmetic in the reference's operation order, no fused multiply-adds (bit-identical to the reference's formulas)",
                       "rays_per_frame": rays, "primary": cnt["primary"], "shadow": cnt["shadow"],
                       "tile_sharding": ("single GPU" if not multi else
                                         f"8x8 tiles, per-GPU lists of equal measured cost (box + triangle tests of a counter frame), gathered to GPU0 each frame"
                                         if balance else f"8x8 tiles, tile t on GPU t % {world}, gathered to GPU0 each frame"),
                       "tile_balance": balance,
                       "gather": None if not multi else ("tiles packed / assembled by the library's kernels (rt_tiles_pack / rt_tiles_scatter)" if use_kernels
                                                         else "tiles packed / assembled by torch index arithmetic"),
                       "frames_in_flight": nf,
                       "frame_latency_ms_with_frames_in_flight": in_flight_stage[0],
                       "fewer_frames_in_flight": sweep,
                       "in_order": {"ms_per_frame": frame_ms, "wall_ms_per_frame": alone_wall, "Mrays_per_s": rays / alone_wall / 1e3,
                                    "note": "one frame in flight: every frame starts after the previous one has finished, as the reference's "
                                            "cudamain + cudaDeviceSynchronize does; `value` overlaps consecutive frames instead"},
                       "default_hw_queues": default_queues,
                       "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "gathered_frame": gather_check,
                       "scene_build_s": round(t_build, 2),
                       "other_workloads": others},
            "kernel_ms": {"frame": frame_ms, "trace_closest": closest_ms, "shade": shade_ms, "trace_shadow": shadow_ms, "rest": rest_ms,
                          "note": "launches running alone (one frame in flight), HIP events on the launch stream, measured after the timed region"},
            "kernel_ms_in_flight": dict(zip(("frame", "trace_closest", "shade", "trace_shadow", "rest"), in_flight_stage)),
            # The dominant kernel is the closest-hit trace of the primary rays.  What bounds it is fp32 vector work and the
            # latency of its dependent loads, not HBM: its working set (BVH + triangle records) stays in L2 / Infinity Cache.
            "roofline": {"bound": "valu_fp32", "kernel": "traceKernel<false,false,false,false> (closest hit, primary rays)",
                         "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_VECTOR_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": closest_flops // world,
                         "launch_ms": closest_ms,
                         "traffic": prof.get("hbm_bytes_per_launch"),
                         "traffic_source": prof_src if prof else None,
                         "note": f"algorithmic flops = {NODE_FLOPS}/box test + {TRI_FLOPS}/triangle test + {SPHERE_FLOPS}/sphere test + "
                                 f"{MESH_ENTRY_FLOPS}/mesh entry, test counts exact from the counter build of the same kernels in this run; "
                                 "duration = average of 10 launches running alone (one frame in flight), HIP events on the launch stream; "
                                 "peak = 256 CUs x 4 SIMDs x 32 lanes x 2 (FMA) x 2.4 GHz. The kernels may not fuse multiply-adds (bit-exactness), so half the "
                                 "peak is out of reach by construction; selects, compares, address and stack work are not counted. The "
                                 "counted tests are the reference walk's; the timed launch fetches four grandchildren per node step and tests "
                                 "boxes the reference walk skips, which the algorithmic count leaves out",
                         "tests_per_ray": {"closest_box": closest["node"] / max(traced, 1), "closest_triangle": closest["tri"] / max(traced, 1),
                                           "shadow_box": sh["node"] / max(cnt["shadow"], 1), "shadow_triangle": sh["tri"] / max(cnt["shadow"], 1)},
                         "hbm": {"bound": "hbm", "achieved": compulsory / (closest_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": compulsory / (closest_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "compulsory_bytes_per_launch": compulsory,
                                 "note": "compulsory bytes = BVH nodes (32 B) + triangle records (48 B) of the scene, read once, + 20 B hit "
                                         "record per ray; `traffic` above is what crossed the L2 -> fabric boundary per launch (PMC: 2 x "
                                         "FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) - L2 capacity misses of a 56 MB working "
                                         "set in 8 x 4 MB L2s, not HBM reads"},
                         "lane_use": wave_statistics(raw),
                         "lane_use_note": "from the counter build, which walks sibling pairs test for test like the reference; the timed launches "
                                          "walk quads (about half as many node steps, each testing four boxes)",
                         "instructions": {"valu_per_launch": prof.get("valu_insts_per_launch"),
                                          "valu_per_box_pair_or_triangle_phase": None,
                                          "source": prof_src if prof else None},
                         "factors": roofline_factors(prof, closest_flops // world),
                         "other_kernels": {"traceKernel<true,false,false,false> (shadow, any hit)": {"achieved": tflops(shadow_flops, shadow_ms), "unit": "TFLOP/s",
                                                                                         "algorithmic_flops_per_launch": shadow_flops // world,
                                                                                         "launch_ms": shadow_ms},
                                           "whole frame, by wall clock, frames in flight": {"achieved": tflops(closest_flops + shadow_flops, ms_per_step),
                                                                                           "unit": "TFLOP/s"}}},
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
