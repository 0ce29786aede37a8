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
