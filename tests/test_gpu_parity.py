"""GPU parity: the HIP render path (through the C-ABI) against the CPU oracle
on the same seeded scenes."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity

pytestmark = pytest.mark.gpu

CASES = [
    # name, w, h, size, lights, depth, tight
    ("cubes", 160, 120, 16, 4, 5, False),
    ("cubes", 160, 120, 16, 4, 5, True),
    ("soup", 160, 96, 20000, 1, 0, False),
    ("soup", 192, 108, 200000, 1, 0, True),
    ("soup", 160, 96, 50000, 10, 0, True),
    ("heightfield", 160, 96, 20000, 2, 0, True),
    ("mixed", 160, 120, 0, 16, 4, False),
    ("mixed", 160, 120, 0, 4, 5, True),
    ("mixed", 161, 67, 0, 2, 0, True),  # ragged tiles, depth 0 with mirror / glass present
]


@pytest.mark.parametrize("one_launch", [False, True])
@pytest.mark.parametrize("name,w,h,size,lights,depth,tight", CASES)
def test_frame_matches_oracle(gpu_ctx, oracle, name, w, h, size, lights, depth, tight, one_launch):
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    gpu_ctx.set_box_test_split(2**31 - 1 if one_launch else 0)  # literal box test in one launch / two launches
    try:
        rgba, keys, cnt, rgba2, keys2 = _frames(gpu_ctx, scene)
    finally:
        gpu_ctx.set_box_test_split(0)
    r = assert_parity(rgba, keys, ref_rgba, ref_keys)
    print(name, r)
    # the counter build walks exactly the oracle's traversal: integer-exact test counts
    assert cnt == ref_cnt, (cnt, ref_cnt)
    # and the plain build renders the same frame as the counter build
    assert np.array_equal(keys, keys2)
    assert np.array_equal(rgba, rgba2, equal_nan=True)


def _frames(ctx, scene):
    rgba, keys, cnt = ctx.render_frame(scene, want_keys=True, want_counters=True)
    rgba2, keys2, _ = ctx.render_frame(scene, want_keys=True)
    return rgba, keys, cnt, rgba2, keys2


@pytest.mark.parametrize("name,w,h,size,lights,depth", [("soup", 192, 108, 200000, 1, 0), ("heightfield", 160, 96, 20000, 2, 0), ("mixed", 160, 120, 0, 4, 3)])
def test_surface_area_tree_frame_matches_oracle(gpu_ctx, oracle, name, w, h, size, lights, depth):
    """Mesh BVHs cut by the binned surface-area heuristic: unbalanced trees, deeper stacks, the same walk on both sides."""
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=True, sah_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt, (cnt, ref_cnt)


@pytest.mark.parametrize("h", [300, 20000])
def test_frame_of_rays_along_an_axis_plane(gpu_ctx, oracle, h):
    """A frame one pixel wide: every primary ray has d.x == 0, so the whole frame is walked by the launch with the
    literal box test - from its list of left-over rays (300 rays), or, when the list is too short (20 000 rays), by
    looking through all of the level's items."""
    scene = rt.HostScene("soup", 1, h, 20000, 2, 0, tight_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, 1, h, threads=8, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    assert (ref_keys[..., 0] >= 0).sum() > h // 20  # the strip does see triangles
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt
    rgba2, keys2, _ = gpu_ctx.render_frame(scene, want_keys=True)
    assert np.array_equal(keys, keys2) and np.array_equal(rgba, rgba2, equal_nan=True)


# ---- OBJ + PNG assets: reader, decoder, tangents, CLAMP and WRAP texturing of 1024x1024 maps -----------------
@pytest.mark.parametrize("name,w,h,lights,depth", [("barrel", 200, 150, 4, 2), ("mixed", 160, 120, 10, 3), ("barrel", 97, 61, 1, 0)])
def test_asset_scenes_match_oracle(gpu_ctx, oracle, synthetic_assets, name, w, h, lights, depth):
    """The scenes of RayTracer::initScene with file assets (cudaTracer/RayTracer.cpp:134-186): a 488-triangle OBJ barrel
    with its three maps, instanced with diffuse and mirror materials, on the WRAP-textured plane, with the glass ball."""
    scene = rt.HostScene(name, w, h, 0, lights, depth, asset_dir=synthetic_assets)
    assert scene.triangles == 3 * 488 + 2
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt, (cnt, ref_cnt)
    assert cnt["shade_mesh"] > 0.2 * w * h and cnt["tri"] > 0


# ---- golden frames (no oracle run involved) -------------------------------------------------
import os  # noqa: E402
import sys  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402


@pytest.mark.parametrize("key", sorted(make_golden.FRAMES))
def test_frame_matches_golden(gpu_ctx, key):
    name, w, h, size, lights, depth, tight = make_golden.FRAMES[key]
    g = np.load(os.path.join(GOLDEN, f"frame_{key}.npz"))
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    assert_parity(rgba, keys, g["rgba"], g["keys"])
    assert [cnt[n] for n in rt.RT_COUNTER_NAMES] == g["counters"].tolist()


# ---- the reference's option switches ---------------------------------------------------------
@pytest.mark.parametrize("scene_bvh,mesh_bvh", [(0, 1), (1, 0), (0, 0)])
def test_linear_scan_modes(gpu_ctx, oracle, scene_bvh, mesh_bvh):
    """Options.cuh:23-24 useSceneBvh / useMeshBvhs off: linear scans over shapes / triangles."""
    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 3, 3, tight_bvh=True)
    scene.opts.contents.useSceneBvh = scene_bvh
    scene.opts.contents.useMeshBvhs = mesh_bvh
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt, (cnt, ref_cnt)


def test_gamma_and_background_options(gpu_ctx, oracle):
    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 2, 2)
    scene.opts.contents.gammaCorrection = 0
    scene.opts.contents.backgroundColor[:] = [0.9, 0.1, 0.3]
    scene.opts.contents.shadowBias = 1e-2
    ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, _ = gpu_ctx.render_frame(scene, want_keys=True)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert np.array_equal(rgba[keys[..., 0] < 0][:, :3], np.broadcast_to(np.float32([0.9, 0.1, 0.3]), (int((keys[..., 0] < 0).sum()), 3)))


# ---- several samples per pixel ----------------------------------------------------------------
@pytest.mark.parametrize("name,samples,depth,w,h", [("mixed", 4, 3, 96, 64), ("soup", 3, 0, 101, 50), ("cubes", 16, 2, 64, 48), ("mixed", 64, 1, 33, 17)])
def test_multi_sample_frames(gpu_ctx, oracle, name, samples, depth, w, h):
    """samples > 1: the jitter is rt_sample_jitter on both sides (the reference's curand draw is not reproducible), the
    samples are summed in order and divided as in Main.cu:66-82; the hit key is sample 0's."""
    scene = rt.HostScene(name, w, h, 20000 if name == "soup" else 6, 3, depth, tight_bvh=True)
    scene.opts.contents.samples = samples
    for seed in (1234, 99):
        gpu_ctx.set_sample_seed(seed)
        oracle.load().oracle_set_sample_seed(seed)
        try:
            ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
            rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
        finally:
            gpu_ctx.set_sample_seed(1234)
            oracle.load().oracle_set_sample_seed(1234)
        assert_parity(rgba, keys, ref_rgba, ref_keys)
        assert cnt == ref_cnt
        assert cnt["primary"] == w * h * samples
    # more samples change the image (edges get intermediate values), one sample is the unjittered frame
    scene.opts.contents.samples = 1
    single = gpu_ctx.render_frame(scene)[0]
    assert not np.array_equal(single, rgba)


@pytest.mark.skipif(False, reason="")
def test_jitter_is_a_centred_uniform():
    import ctypes as C

    lib = rt.load()
    # the header's inline function, through the oracle-independent C harness is not exported; recompute it here
    def h32(v):
        v &= 0xFFFFFFFF
        state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return ((word >> 22) ^ word) & 0xFFFFFFFF

    def jitter(seed, pixel, sample, axis):
        h = h32((h32(pixel ^ seed) + (sample * 2 + axis) * 0x9E3779B9) & 0xFFFFFFFF)
        return (h >> 8) / 16777216.0 - 0.5

    vals = np.array([jitter(1234, p, s, a) for p in range(2000) for s in range(4) for a in range(2)])
    assert vals.min() >= -0.5 and vals.max() < 0.5
    assert abs(vals.mean()) < 0.01 and abs(vals.std() - (1 / 12) ** 0.5) < 0.01
    assert lib is not None


# ---- kernel knobs must not change a single bit -----------------------------------------------
def test_knobs_do_not_change_the_frame(gpu_ctx):
    scene = rt.HostScene("soup", 192, 108, 100000, 2, 0, tight_bvh=True)
    dscene = gpu_ctx.upload(scene)
    try:
        base = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        for setter, values, restore in ((gpu_ctx.set_lds_stack, (1, 3, 8, 64), 8), (gpu_ctx.set_frames_in_flight, (2, 4), 1), (gpu_ctx.set_tile_order, (0, 1), 1),
                                        (gpu_ctx.set_trace_waves, (1, 4, 32), 0), (gpu_ctx.set_box_test_split, (2**31 - 1, 20000), 0)):
            try:
                for v in values:
                    setter(v)
                    got = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
                    assert np.array_equal(got[1], base[1]), (setter.__name__, v)
                    assert np.array_equal(got[0], base[0], equal_nan=True), (setter.__name__, v)
                    assert got[2] == base[2], (setter.__name__, v)
                    plain = gpu_ctx.render_frame(scene, dscene, want_keys=True)  # the build without counters
                    assert np.array_equal(plain[1], base[1]) and np.array_equal(plain[0], base[0], equal_nan=True), (setter.__name__, v)
            finally:
                setter(restore)
    finally:
        gpu_ctx.free_scene(dscene)


def test_deep_stack_spills_to_global_memory(gpu_ctx, oracle):
    """A 1-entry LDS stack forces nearly every push through the spill buffer, with secondary rays in flight."""
    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 3, 4)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    gpu_ctx.set_lds_stack(1)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
        # the plain build walks quads (up to three pushes per node step) where the counter build walks pairs
        rgba2, keys2, _ = gpu_ctx.render_frame(scene, want_keys=True)
    finally:
        gpu_ctx.set_lds_stack(8)
    assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert cnt == ref_cnt
    assert np.array_equal(keys, keys2)
    assert np.array_equal(rgba, rgba2, equal_nan=True)


def test_overlapping_boxes_fill_the_stack_of_the_quad_walk(gpu_ctx, oracle):
    """Triangles stacked along the view direction in a deep tree whose boxes all hold the centre rays: every box test
    passes, so the walk over quads keeps three entries waiting per step, well past the LDS ring."""
    import hand_scenes as hs
    from test_gpu_fuzz import bvh_of

    import cudaraytracer_amd

    rtlib = cudaraytracer_amd.load()
    n = 3000
    tris = []
    for k in range(n):
        z = 4.0 + 0.01 * k
        s = 3.0 + 0.001 * (k % 7)
        tris.append(hs.tri((-s, -s, z), (-s, 2 * s, z), (2 * s, -s, z)) if k % 3 else hs.tri((-s, -s, z), (2 * s, -s, z), (-s, 2 * s, z)))
    tris, spec = bvh_of(rtlib, tris, 0, 1, 1)  # median cut down to single triangles: depth 12, every box around the axis
    scene = hs.Scene(17, 13, [hs.Mesh(tris, nodes=spec)], lights=[((0, 0, 0), (1, 1, 1), 5.0), ((1, 2, 0), (1, 1, 1), 3.0)])
    ref_rgba, ref_keys = scene.render_oracle(oracle)
    gpu_ctx.set_lds_stack(2)
    try:
        rgba, keys = scene.render_gpu(gpu_ctx)
    finally:
        gpu_ctx.set_lds_stack(8)
    assert (ref_keys[..., 0] == 0).mean() > 0.3
    assert np.array_equal(keys, ref_keys)
    assert np.array_equal(rgba, ref_rgba, equal_nan=True)


# ---- tile shards, scene hand-over, lights ------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_shards_compose_the_frame(gpu_ctx, world):
    w, h = 200, 120
    scene = rt.HostScene("mixed", w, h, 0, 3, 3)
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(w, h)
    keys = gpu_ctx.alloc(w * h * 16)
    cnt = gpu_ctx.alloc(rt.RT_COUNTER_CNT * 8)
    try:
        full = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        gpu_ctx.zero(surf, pitch * h)
        gpu_ctx.zero(keys, w * h * 16)
        gpu_ctx.zero(cnt, rt.RT_COUNTER_CNT * 8)
        for r in range(world):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys, cnt, r, world)
        gpu_ctx.sync()
        rgba = gpu_ctx.download_surface(surf, pitch, w, h)
        k = gpu_ctx.download(keys, (h, w, 4), np.int32)
        c = gpu_ctx.download(cnt, (rt.RT_COUNTER_CNT,), np.uint64)
        assert np.array_equal(k, full[1])
        assert np.array_equal(rgba, full[0], equal_nan=True)
        assert {n: int(c[i]) for i, n in enumerate(rt.RT_COUNTER_NAMES)} == full[2]  # counters add up over shards
    finally:
        gpu_ctx.free(surf)
        gpu_ctx.free(keys)
        gpu_ctx.free(cnt)
        gpu_ctx.free_scene(dscene)


@pytest.mark.parametrize("world", [1, 3])
def test_cost_balanced_tile_lists_compose_the_frame(gpu_ctx, world):
    """A counter frame books every tile's box + triangle tests; balanced_assignment turns them into per-rank tile lists
    of equal cost; rendered list by list (rt_set_tile_list) the frame is the one rendered whole, and the lists' measured
    costs are closer together than those of the interleaved shards."""
    from cudaraytracer_amd import multi as mg

    w, h = 203, 117  # ragged tiles
    scene = rt.HostScene("soup", w, h, 30000, 2, 0, tight_bvh=True)
    dscene = gpu_ctx.upload(scene)
    tx, ty = mg.tile_grid(w, h)
    surf, pitch = gpu_ctx.surface(w, h)
    keys = gpu_ctx.alloc(w * h * 16)
    cnt = gpu_ctx.alloc(rt.RT_COUNTER_CNT * 8)
    cost = gpu_ctx.alloc(tx * ty * 4)
    try:
        gpu_ctx.set_tile_cost_buffer(cost)
        full = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        costs = gpu_ctx.download(cost, (tx * ty,), np.uint32).astype(np.int64)
        # depth 0: every ray is a primary ray or the shadow ray of a primary hit, so the tiles' costs add up to the frame's tests
        assert int(costs.sum()) == full[2]["node"] + full[2]["tri"]
        assert costs.min() > 0 and costs.max() > 3 * np.median(costs)  # the soup's centre is far more expensive than its rim
        lists = mg.balanced_assignment(costs, world)
        spread = lambda ls: max(int(costs[np.asarray(l)].sum()) for l in ls) - min(int(costs[np.asarray(l)].sum()) for l in ls)  # noqa: E731
        if world > 1:
            assert spread(lists) <= spread(mg.interleaved_assignment(w, h, world))
        gpu_ctx.set_tile_cost_buffer(None)
        gpu_ctx.zero(surf, pitch * h)
        gpu_ctx.zero(keys, w * h * 16)
        gpu_ctx.zero(cnt, rt.RT_COUNTER_CNT * 8)
        for l in lists:
            gpu_ctx.set_tile_list(l.numpy())
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys, cnt)
        gpu_ctx.sync()
        assert np.array_equal(gpu_ctx.download(keys, (h, w, 4), np.int32), full[1])
        assert np.array_equal(gpu_ctx.download_surface(surf, pitch, w, h), full[0], equal_nan=True)
        c = gpu_ctx.download(cnt, (rt.RT_COUNTER_CNT,), np.uint64)
        assert {n: int(c[i]) for i, n in enumerate(rt.RT_COUNTER_NAMES)} == full[2]
        # a list and an arithmetic shard in one call, a tile outside the frame, a tile twice: refused
        with pytest.raises(rt.RtError, match="not both"):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, None, None, 1, 2)
        gpu_ctx.set_tile_list([0, tx * ty])
        with pytest.raises(rt.RtError, match="tiles"):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        gpu_ctx.set_tile_list([2, 7])
        with pytest.raises(rt.RtError, match="twice"):
            gpu_ctx.set_tile_list([3, 5, 3])
        with pytest.raises(rt.RtError, match="negative"):
            gpu_ctx.set_tile_list([4, -1])
        # a rejected list leaves the current one in place: the next call renders tiles 2 and 7 and nothing else
        gpu_ctx.zero(surf, pitch * h)
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        gpu_ctx.sync()
        got = gpu_ctx.download_surface(surf, pitch, w, h)
        drawn = np.zeros((h, w), dtype=bool)
        for t in (2, 7):
            drawn[(t // tx) * 8:(t // tx) * 8 + 8, (t % tx) * 8:(t % tx) * 8 + 8] = True
        assert np.array_equal(got[drawn], full[0][drawn], equal_nan=True) and (got[~drawn] == 0).all()
    finally:
        gpu_ctx.set_tile_list(None)
        gpu_ctx.set_tile_cost_buffer(None)
        for p in (surf, keys, cnt, cost):
            gpu_ctx.free(p)
        gpu_ctx.free_scene(dscene)


@pytest.mark.parametrize("in_flight,frames", [(3, 6), (16, 40)])
def test_frames_in_flight_overlap_without_mixing(gpu_ctx, in_flight, frames):
    """Several frames in flight, frames of two different scenes into their own surfaces (twice round the slots and
    more): each equals its in-order render."""
    w, h = 160, 96
    scenes = [rt.HostScene("mixed", w, h, 0, 3, 3), rt.HostScene("soup", w, h, 30000, 2, 0, tight_bvh=True)]
    dscenes = [gpu_ctx.upload(s) for s in scenes]
    expect = [gpu_ctx.render_frame(s, d, want_keys=True) for s, d in zip(scenes, dscenes)]
    surfs = [gpu_ctx.surface(w, h) for _ in range(frames)]
    keys = [gpu_ctx.alloc(w * h * 16) for _ in range(frames)]
    with pytest.raises(rt.RtError):
        gpu_ctx.set_frames_in_flight(17)
    gpu_ctx.set_frames_in_flight(in_flight)
    try:
        for i in range(frames):
            gpu_ctx.render(dscenes[i % 2], scenes[i % 2].opts, scenes[i % 2].camera, surfs[i][0], surfs[i][1], keys[i])
        # a stream can be ordered behind one frame without a host wait
        gpu_ctx.stream_wait_frame(None, 0)
        gpu_ctx.stream_wait_frame(None, 2)
        gpu_ctx.sync()
        for i in range(frames):
            rgba = gpu_ctx.download_surface(surfs[i][0], surfs[i][1], w, h)
            k = gpu_ctx.download(keys[i], (h, w, 4), np.int32)
            assert np.array_equal(k, expect[i % 2][1]), i
            assert np.array_equal(rgba, expect[i % 2][0], equal_nan=True), i
    finally:
        gpu_ctx.set_frames_in_flight(1)
        for sf, k in zip(surfs, keys):
            gpu_ctx.free(sf[0])
            gpu_ctx.free(k)
        for d in dscenes:
            gpu_ctx.free_scene(d)


def test_adopted_blob_renders_like_an_uploaded_scene(gpu_ctx):
    import torch

    w, h = 128, 80
    scene = rt.HostScene("mixed", w, h, 0, 2, 3)
    base = gpu_ctx.render_frame(scene, want_keys=True)
    blob = torch.from_numpy(gpu_ctx.pack(scene.desc)).to("cuda:0")
    torch.cuda.synchronize()
    dscene = gpu_ctx.adopt_blob(blob.data_ptr(), blob.numel())
    try:
        got = gpu_ctx.render_frame(scene, dscene, want_keys=True)
        assert np.array_equal(got[1], base[1]) and np.array_equal(got[0], base[0], equal_nan=True)
        # and the blob read back from an uploaded scene is the packed blob
        up = gpu_ctx.upload(scene)
        ptr, n = gpu_ctx.scene_blob(up)
        assert n == blob.numel()
        assert np.array_equal(gpu_ctx.download(ptr, (n,), np.uint8), blob.cpu().numpy())
        gpu_ctx.free_scene(up)
    finally:
        gpu_ctx.free_scene(dscene)
    with pytest.raises(rt.RtError):
        gpu_ctx.adopt_blob(blob.data_ptr() + 256, blob.numel() - 256)  # not a blob header


def test_moving_lights_without_a_new_upload(gpu_ctx, oracle):
    """RayTracer.cpp:199-205 rotates the lights every frame and re-uploads only the light table."""
    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 3, 2)
    dscene = gpu_ctx.upload(scene)
    try:
        for _ in range(2):
            scene.rotate_lights(0.7)
            gpu_ctx.set_lights(dscene, scene.desc)
            ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
            rgba, keys, _ = gpu_ctx.render_frame(scene, dscene, want_keys=True)
            assert_parity(rgba, keys, ref_rgba, ref_keys)
    finally:
        gpu_ctx.free_scene(dscene)


def test_path_budget_overflow_is_reported(gpu_ctx):
    """Glass spawns two children per hit; with one node per pixel the arrays overflow and rt_sync says so."""
    scene = rt.HostScene("mixed", 128, 96, 0, 2, 6)
    gpu_ctx.set_path_budget(1)
    try:
        with pytest.raises(rt.RtError, match="budget|overflow|path"):
            gpu_ctx.render_frame(scene)
    finally:
        gpu_ctx.set_path_budget(3)
    gpu_ctx.render_frame(scene)  # the context stays usable


def test_path_budget_overflow_stays_inside_the_workspace(oracle):
    """Budget 1 with glass at depth 6: every level asks for more nodes than the arrays hold.  No array may be written
    past its end (guard bands behind each one stay intact), the nodes that fit are traced like the oracle's, and the
    flag survives later frames on the same slot until rt_sync reports it."""
    w, h = 128, 96
    scene = rt.HostScene("mixed", w, h, 0, 2, 6)
    ok = rt.HostScene("cubes", w, h, 3, 1, 0)
    ctx = rt.Context(0)
    try:
        ctx.set_workspace_guards(4096)
        ctx.set_path_budget(1)
        dscene, dok = ctx.upload(scene), ctx.upload(ok)
        surf, pitch = ctx.surface(w, h)
        for frames in (1, 3):
            ctx.set_frames_in_flight(frames)
            for _ in range(frames):
                ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
            # frames that do not overflow, on the same slots, before anybody syncs
            for _ in range(2 * frames):
                ctx.render(dok, ok.opts, ok.camera, surf, pitch)
            with pytest.raises(rt.RtError, match="budget"):
                ctx.sync()
            ctx.sync()  # reported once
            ctx.check_workspace_guards()
        # level 0 fits: pixels whose primary hit is diffuse do not depend on the dropped children
        ctx.set_frames_in_flight(1)
        keys = ctx.alloc(w * h * 16)
        ctx.render(dscene, scene.opts, scene.camera, surf, pitch, keys)
        with pytest.raises(rt.RtError, match="budget"):
            ctx.sync()
        rgba = ctx.download_surface(surf, pitch, w, h)
        got_keys = ctx.download(keys, (h, w, 4), np.int32)
        ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
        assert (got_keys == ref_keys).all()
        from parity import compare

        assert compare(rgba, None, ref_rgba, None)["radiance_ok_fraction"] > 0.5  # the diffuse part of the image
        ctx.check_workspace_guards()
    finally:
        ctx.close()


def test_frame_limits_are_checked_before_launch(gpu_ctx):
    """Shadow work items are (hit, light) pairs in 32-bit indices: a frame whose path nodes x lights pass 2^31 is
    refused.  (The surface is large enough for the frame, so the call is valid in every other respect.)"""
    w = h = 2048
    scene = rt.HostScene("soup", w, h, 1000, 16, 0)
    assert scene.desc.contents.lightCnt == 16
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(w, h)
    scene.opts.contents.samples = 64  # 2^28 path nodes x 16 lights = 2^32
    try:
        with pytest.raises(rt.RtError, match="32-bit shadow-ray indices"):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
    finally:
        gpu_ctx.free(surf)
        gpu_ctx.free_scene(dscene)


def test_empty_tile_shard_counts_as_a_frame(gpu_ctx):
    """More ranks than tiles: a call without tiles launches nothing but keeps rt_stream_wait_frame's frame count."""
    scene = rt.HostScene("cubes", 16, 8, 2, 1, 0)  # 2 tiles
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(16, 8)
    try:
        gpu_ctx.set_frames_in_flight(2)
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, None, None, 1, 4)  # tile 1
        gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, None, None, 3, 4)  # no tile
        gpu_ctx.stream_wait_frame(None, 0)
        gpu_ctx.stream_wait_frame(None, 1)
        gpu_ctx.sync()
    finally:
        gpu_ctx.set_frames_in_flight(1)
        gpu_ctx.free(surf)
        gpu_ctx.free_scene(dscene)


def test_bad_render_arguments(gpu_ctx):
    scene = rt.HostScene("cubes", 64, 48, 3, 1, 0)
    dscene = gpu_ctx.upload(scene)
    surf, pitch = gpu_ctx.surface(64, 48)
    try:
        with pytest.raises(rt.RtError):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, 64 * 16 - 4)  # pitch too small
        with pytest.raises(rt.RtError):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch, None, None, 3, 2)  # tileFirst >= tileStride
        scene.opts.contents.maxDepth = 99
        with pytest.raises(rt.RtError):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        scene.opts.contents.maxDepth = 0
        scene.opts.contents.samples = 0
        with pytest.raises(rt.RtError):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        scene.opts.contents.samples = 65
        with pytest.raises(rt.RtError):
            gpu_ctx.render(dscene, scene.opts, scene.camera, surf, pitch)
        with pytest.raises(rt.RtError):
            gpu_ctx.set_box_test_split(-1)
    finally:
        gpu_ctx.free(surf)
        gpu_ctx.free_scene(dscene)


# ---- the headline workload at full size ------------------------------------------------------
def test_full_size_soup_frame_and_properties(gpu_ctx, oracle):
    """The bench workload itself, 1080p over the 1M-triangle soup: the whole frame and its test counts against the
    oracle, and size-independent properties (determinism, ray-count identities)."""
    w, h = 1920, 1080
    scene = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=True)
    dscene = gpu_ctx.upload(scene)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=16, shadow_any_hit=True)
        r = assert_parity(rgba, keys, ref_rgba, ref_keys)
        assert r["key_mismatches"] == 0 and cnt == ref_cnt, (r, cnt, ref_cnt)
        hit = keys[..., 0] >= 0
        assert cnt["primary"] == w * h
        assert cnt["shade_mesh"] == int(hit.sum()) and cnt["shadow"] == int(hit.sum()) * 1
        assert 0.3 < hit.mean() < 0.9
        again = gpu_ctx.render_frame(scene, dscene, want_keys=True)
        assert np.array_equal(again[1], keys) and np.array_equal(again[0], rgba, equal_nan=True)
    finally:
        gpu_ctx.free_scene(dscene)


def test_headline_scene_on_its_own_tree_at_full_size(gpu_ctx, oracle):
    """The bench's headline exactly: 1080p over the 1M-triangle soup on the tight-bounds tree cut by the binned
    surface-area heuristic (unbalanced, deeper than the median-cut tree above).  Whole frame against the oracle: every
    hit key, the radiance, and integer-equal test counters."""
    w, h = 1920, 1080
    scene = rt.HostScene("soup", w, h, 1000000, 1, 0, tight_bvh=True, sah_bvh=True)
    dscene = gpu_ctx.upload(scene)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=16, shadow_any_hit=True)
        r = assert_parity(rgba, keys, ref_rgba, ref_keys)
        assert r["key_mismatches"] == 0 and cnt == ref_cnt, (r, cnt, ref_cnt)
        assert cnt["primary"] == w * h and cnt["shadow"] == int((keys[..., 0] >= 0).sum())
        # the plain build (quad walk, two launches per trace) renders the counter build's frame
        plain = gpu_ctx.render_frame(scene, dscene, want_keys=True)
        assert np.array_equal(plain[1], keys) and np.array_equal(plain[0], rgba, equal_nan=True)
    finally:
        gpu_ctx.free_scene(dscene)


def test_ten_million_triangle_heightfield_at_4k(gpu_ctx, oracle):
    """The largest configuration: 9 999 392 triangles in one mesh BVH (depth 20, a 2.2 GB scene blob), 3840x2160.
    Whole frame, hit keys and test counts against the oracle."""
    w, h = 3840, 2160
    scene = rt.HostScene("heightfield", w, h, 0, 1, 0, tight_bvh=True)
    assert scene.triangles == 9999392 and scene.bvh_depth() == (0, 20)
    dscene = gpu_ctx.upload(scene)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=16, shadow_any_hit=True)
        r = assert_parity(rgba, keys, ref_rgba, ref_keys)
        assert r["key_mismatches"] == 0 and cnt == ref_cnt, (r, cnt, ref_cnt)
        assert cnt["primary"] == w * h and cnt["shadow"] == int((keys[..., 0] >= 0).sum())
    finally:
        gpu_ctx.free_scene(dscene)


@pytest.mark.parametrize("label,name,w,h,size,lights,depth", [
    ("cfg 1: 16x16 cube meshes under a scene BVH", "cubes", 640, 480, 16, 4, 5),
    ("cfg 3: barrel instances, textured plane, mirror and glass balls", "barrel", 1920, 1080, 0, 10, 2),
    ("cfg 5: default-scene analogue", "mixed", 1920, 1080, 0, 16, 4),
])
def test_other_workloads_at_full_size(gpu_ctx, oracle, synthetic_assets, label, name, w, h, size, lights, depth):
    """The bench's other workloads at the sizes it runs them (mesh and texture files: stand-ins of the reference's
    shape): the whole frame against the oracle - every hit key, the radiance with its NaN pixels, integer-equal test and
    ray counters through all reflection levels - and the plain build's frame against the counter build's."""
    scene = rt.HostScene(name, w, h, size, lights, depth, asset_dir=None if name == "cubes" else synthetic_assets, tight_bvh=True)
    dscene = gpu_ctx.upload(scene)
    try:
        rgba, keys, cnt = gpu_ctx.render_frame(scene, dscene, want_keys=True, want_counters=True)
        ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=16, shadow_any_hit=True)
        r = assert_parity(rgba, keys, ref_rgba, ref_keys)
        assert r["key_mismatches"] == 0 and cnt == ref_cnt, (label, r, cnt, ref_cnt)
        assert cnt["primary"] == w * h
        if depth > 0 and name != "cubes":
            assert cnt["secondary"] > 0  # the mirror and glass materials did spawn rays
        plain = gpu_ctx.render_frame(scene, dscene, want_keys=True)
        assert np.array_equal(plain[1], keys) and np.array_equal(plain[0], rgba, equal_nan=True)
    finally:
        gpu_ctx.free_scene(dscene)


@pytest.mark.parametrize("name,size,lights,depth", [("soup", 50000, 2, 0), ("mixed", 0, 2, 2)])
def test_8k_frame(gpu_ctx, oracle, name, size, lights, depth):
    """7680x4320: 33 M pixels, with the mixed scene up to 100 M path nodes - index arithmetic at its largest.  Rows
    sampled across the frame against the oracle, ray-count identities over the whole frame."""
    w, h = 7680, 4320
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    rows = (7, h, 431)
    ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, rows=rows, threads=16, shadow_any_hit=True)
    sel = slice(*rows)
    r = assert_parity(rgba[sel], keys[sel], ref_rgba[sel], ref_keys[sel])
    assert r["key_mismatches"] == 0
    assert cnt["primary"] == w * h and cnt["shade_mesh"] + cnt["shade_sphere"] >= int((keys[..., 0] >= 0).sum())
    assert np.all(rgba[..., 3] == 1.0)


def test_cpp_driver_frame_loop_and_png_dump(oracle, tmp_path):
    """RayTracer.h (the C++ mirror of cudaTracer/RayTracer.cpp) drives three frames with moving lights through the
    C-ABI and dumps the last one; the oracle renders the same final state."""
    from PIL import Image

    from parity import to_8bit

    w, h = 96, 64
    scene = rt.HostScene("mixed", w, h, 0, 3, 2)
    png = tmp_path / "frame.png"
    rgba = scene.run_frames(3, 0.4, str(png))
    ref_rgba, ref_keys, _ = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8)  # lights as after 3 rotations
    assert_parity(rgba, None, ref_rgba, None)
    other = rt.HostScene("mixed", w, h, 0, 3, 2)
    still = oracle.render(other.desc, other.opts, other.camera, w, h, threads=8)[0]
    assert not np.array_equal(still, ref_rgba)  # the lights did move
    dumped = np.asarray(Image.open(png).convert("RGBA")).astype(np.int32)
    assert np.abs(dumped - to_8bit(ref_rgba)).max() <= 1
