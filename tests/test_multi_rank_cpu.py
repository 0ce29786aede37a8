"""world_size-2 (and 5) runs of the multi-GPU frame logic on the gloo backend:
scene sharing, interleaved tile assignment and the gather to rank 0.  The
per-rank renderer is stood in for by the CPU oracle (each rank traces the frame
and keeps only its tiles), so the assembled frame must equal a single-process
frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H = 64, 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cudaraytracer_amd as rt
        from cudaraytracer_amd import multi as mg
        from oracle import pyoracle

        dev = torch.device("cpu")
        blob0 = opts0 = cam0 = None
        scene = rt.HostScene("mixed", W, H, 0, 2, 2)  # deterministic: every rank can rebuild it for the stand-in renderer
        if rank == 0:
            lib = rt.load()
            blob0 = torch.from_numpy(rt.pack_scene(lib, scene.desc))
            opts0 = bytes(scene.opts.contents)
            cam = scene.camera.contents
            cam0 = list(cam.orig) + list(cam.camera)
        blob, opts_raw, cam_f = mg.share_scene(blob0, opts0, cam0, rank, world, dev)
        # every rank got rank 0's bytes
        assert opts_raw == bytes(scene.opts.contents)
        cam = scene.camera.contents
        assert np.array_equal(cam_f, np.array(list(cam.orig) + list(cam.camera), dtype=np.float32))
        sums = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sums, torch.tensor([blob.numel(), int(blob.to(torch.int64).sum())]))
        assert all(torch.equal(s, sums[0]) for s in sums)

        rgba, _, _ = pyoracle.render(scene.desc, scene.opts, scene.camera, W, H, threads=2)
        full = torch.from_numpy(rgba.copy())
        # a rank only owns its tiles: poison everything else so that a wrong gather shows
        g = mg.TileGather(W, H, rank, world, dev)
        tx, ty = mg.tile_grid(W, H)
        own = torch.zeros(tx * ty, dtype=torch.bool)
        own[mg.tile_ids(W, H, rank, world)] = True
        mask = own.view(ty, tx)[:, None, :, None].expand(ty, 8, tx, 8).reshape(H, W)
        surface = torch.where(mask[:, :, None], full, torch.full_like(full, -777.0))
        frame = g.gather(surface)
        if rank == 0:
            assert torch.equal(frame, full)
            second = g.gather(surface)  # buffers are reused across frames
            assert torch.equal(second, full)
            open(out_path, "w").write("ok")
        else:
            g.gather(surface)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])  # 48 tiles: even split, and a split with a remainder
def test_tile_sharded_frame_equals_single_rank_frame(world, tmp_path):
    out = tmp_path / "ok"
    mp.spawn(_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_tile_assignment_covers_the_frame_once():
    from cudaraytracer_amd import multi as mg

    for (w, h) in [(64, 48), (1920, 1080), (100, 37)]:
        tx, ty = mg.tile_grid(w, h)
        for world in (1, 2, 3, 4, 8):
            ids = torch.cat([mg.tile_ids(w, h, r, world) for r in range(world)])
            assert sorted(ids.tolist()) == list(range(tx * ty))
            counts = [mg.tile_count(w, h, r, world) for r in range(world)]
            assert max(counts) - min(counts) <= 1


# ---- ragged frames, cost-balanced tile lists, 8-bit gather, big-blob sharing, frames in flight -----------------------------
W2, H2 = 100, 37  # 13 x 5 tiles, the last column and row only partly inside the frame


def _frame(i):
    """A frame whose every pixel says which frame it belongs to and where it sits."""
    y, x = torch.meshgrid(torch.arange(H2, dtype=torch.float32), torch.arange(W2, dtype=torch.float32), indexing="ij")
    return torch.stack([x / W2, y / H2, torch.full_like(x, (i % 7) / 8.0), torch.ones_like(x)], dim=2)


def _worker2(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cudaraytracer_amd import multi as mg

        dev = torch.device("cpu")
        tx, ty = mg.tile_grid(W2, H2)
        # a cost table with a hot spot: the balanced lists differ from the interleaved ones but every rank agrees on them
        cost = torch.arange(tx * ty, dtype=torch.float64) % 11 + 1
        cost[:6] = 500
        lists = mg.balanced_assignment(cost, world)
        loads = [float(cost[l].sum()) for l in lists]
        inter = [float(cost[l].sum()) for l in mg.interleaved_assignment(W2, H2, world)]
        assert max(loads) - min(loads) <= max(inter) - min(inter)
        assert max(len(l) for l in lists) - min(len(l) for l in lists) <= 1
        assert all(float(cost[l[0]]) >= float(cost[l[-1]]) for l in lists)  # expensive first

        def poisoned(full, mine):
            own = torch.zeros(tx * ty, dtype=torch.bool)
            own[mine] = True
            mask = own.view(ty, tx)[:, None, :, None].expand(ty, 8, tx, 8).reshape(ty * 8, tx * 8)[:H2, :W2]
            return torch.where(mask[:, :, None], full, torch.full_like(full, -777.0))

        for assignment in (None, lists):
            mine = (lists if assignment is not None else mg.interleaved_assignment(W2, H2, world))[rank]
            g = mg.TileGather(W2, H2, rank, world, dev, assignment=assignment)
            frame = g.gather(poisoned(_frame(3), mine))
            if rank == 0:
                assert torch.equal(frame, _frame(3))
            # the same gatherer with new lists (re-balancing between frames)
            g.set_assignment(mg.interleaved_assignment(W2, H2, world))
            frame = g.gather(poisoned(_frame(4), mg.interleaved_assignment(W2, H2, world)[rank]))
            if rank == 0:
                assert torch.equal(frame, _frame(4))
        # 8-bit gather: the PNG dump's rule
        g8 = mg.TileGather(W2, H2, rank, world, dev, assignment=lists, quantize=True)
        f = _frame(5) * 1.2 - 0.05  # values below 0 and above 1 clamp
        f[0, 0, 0] = float("nan")
        frame8 = g8.gather(poisoned(f, lists[rank]))
        if rank == 0:
            assert frame8.dtype == torch.uint8 and torch.equal(frame8, mg.quantize8(f))
            assert frame8[0, 0, 0] == 0 and int(frame8.max()) == 255

        # scene sharing: broadcast and scatter + all-gather deliver the same bytes, also when the size does not divide
        blob0 = (torch.arange(1000003, dtype=torch.int64) * 7919 % 251).to(torch.uint8) if rank == 0 else None
        want = (torch.arange(1000003, dtype=torch.int64) * 7919 % 251).to(torch.uint8)
        for sg in (False, True):
            got = mg.share_blob(blob0, 1000003, rank, world, dev, scatter_gather=sg)
            assert torch.equal(got, want)

        # frames in flight: 3 surfaces, 8 frames, every gathered frame is the right one
        nf, seen = 3, []
        g = mg.TileGather(W2, H2, rank, world, dev, assignment=lists)
        surfaces = [torch.zeros(H2, W2, 4) for _ in range(nf)]

        def render(surface, i):
            surface.copy_(poisoned(_frame(i), lists[rank]))

        def check(i):
            seen.append(i)
            if rank == 0:
                assert torch.equal(g.assembled(), _frame(i)), f"gathered frame {i} is wrong"

        pipe = mg.FramePipeline(surfaces, render, None, g, None, check)
        for _ in range(8):
            pipe.step()
        pipe.drain()
        assert seen == list(range(8))
        for _ in range(2):  # fewer frames than surfaces
            pipe.step()
        pipe.drain()
        assert seen == list(range(8)) + [0, 1]
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_balanced_lists_ragged_frames_and_frames_in_flight(world, tmp_path):
    out = tmp_path / "ok"
    mp.spawn(_worker2, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_center_out_keeps_the_tiles_and_orders_them_like_the_library():
    """multi.center_out: the same tiles, nearest to the image centre first (the key of rt_capi.cpp's centre-out table)."""
    import torch

    from cudaraytracer_amd import multi as mg

    w, h = 100, 37
    tx, ty = mg.tile_grid(w, h)
    tiles = torch.randperm(tx * ty)[: tx * ty // 2]
    out = mg.center_out(tiles, w, h)
    assert sorted(out.tolist()) == sorted(tiles.tolist())
    key = [((int(t) % tx) * 16 + 8 - w) ** 2 + ((int(t) // tx) * 16 + 8 - h) ** 2 for t in out.tolist()]
    assert key == sorted(key)
    assert mg.center_out([], w, h).numel() == 0
