"""Multi-GPU frame rendering: one process per GPU, the scene blob broadcast once,
every rank renders the 8x8 tiles t = rank, rank + world, ... of the SAME frame and
the tiles are gathered to rank 0.

The reference is single-GPU (cudaTracer/Main.cu:46-90 launches one grid over the
whole surface); the tile of this module is the reference's thread block made
8x8 pixels wide, and the interleaved assignment balances sky / geometry tiles
across GPUs without a work queue between them.

The only collectives are the one-off scene broadcast and the per-frame gather
of finished tiles (batched point-to-point sends to rank 0: every peer has its own xGMI
link to GPU 0, so the N-1 transfers run in parallel).  The functions work on
any torch.distributed backend: "nccl" (RCCL) with device tensors in production,
"gloo" with host tensors in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

TILE = 8
OPTS_BYTES = 48
CAMERA_FLOATS = 3 + 16


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tile_ids(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return torch.arange(rank, tx * ty, world)


def tile_count(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return len(range(rank, tx * ty, world))


# A blob of this size or more travels as scatter + all-gather: rank 0 sends every rank a different 1/N slice and the
# ranks then exchange slices, so that all of a GPU's xGMI links carry a share at once; a plain broadcast moves the
# whole blob over one link after the other (ring) and is bound by a single link's rate.  Small blobs are latency-bound
# either way and take the one-call broadcast.
SCATTER_GATHER_BYTES = 64 << 20


def share_blob(blob, nbytes, rank, world, device, scatter_gather=None):
    """The scene blob (uint8 tensor on rank 0, None elsewhere) on every rank."""
    if world == 1:
        return blob
    if scatter_gather is None:
        scatter_gather = nbytes >= SCATTER_GATHER_BYTES
    if not scatter_gather:
        if rank != 0:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dist.broadcast(blob, 0)
        return blob
    part = (nbytes + world - 1) // world
    part = (part + 255) // 256 * 256  # slices keep the blob's 256-byte section alignment
    out = torch.empty(part * world, dtype=torch.uint8, device=device)
    slices = list(out.split(part))
    if rank == 0:
        out[:nbytes].copy_(blob)
        dist.scatter(slices[0], [s.clone() if i == 0 else s for i, s in enumerate(slices)], src=0)
    else:
        dist.scatter(slices[rank], None, src=0)
    mine = slices[rank].clone()
    dist.all_gather(slices, mine)
    return out[:nbytes]


def share_scene(blob, opts_bytes, camera_floats, rank, world, device, scatter_gather=None):
    """Rank 0 passes the packed scene blob (uint8 tensor on `device`), the 48 raw bytes of DeviceOpts and the
    19 camera floats (origin + matrix); the other ranks pass None.  Returns (blob, opts_bytes, camera_floats) on every rank."""
    head = torch.zeros(1, dtype=torch.int64, device=device)
    small = torch.zeros(OPTS_BYTES + 4 * CAMERA_FLOATS, dtype=torch.uint8, device=device)
    if rank == 0:
        head[0] = blob.numel()
        cam = np.asarray(camera_floats, dtype=np.float32)
        assert len(opts_bytes) == OPTS_BYTES and cam.size == CAMERA_FLOATS
        small.copy_(torch.frombuffer(bytearray(bytes(opts_bytes) + cam.tobytes()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(head, 0)
        dist.broadcast(small, 0)
    blob = share_blob(blob, int(head.item()), rank, world, device, scatter_gather)
    raw = small.cpu().numpy().tobytes()
    return blob, raw[:OPTS_BYTES], np.frombuffer(raw[OPTS_BYTES:], dtype=np.float32).copy()


class TileGather:
    """Collects the tiles each rank rendered into the full frame on rank 0.  Frame sizes must be multiples of 8.

    Rank 0 receives every peer's tiles straight into one [world, tiles per rank, 8, 8, c] buffer (one irecv per peer,
    all in flight together) and interleaves them into tile order with a single strided copy."""

    def __init__(self, width, height, rank, world, device, channels=4, dtype=torch.float32):
        if width % TILE or height % TILE:
            raise ValueError("tile gather needs a frame size divisible by 8")
        self.w, self.h, self.rank, self.world = width, height, rank, world
        self.tx, self.ty = tile_grid(width, height)
        self.c = channels
        self.n = self.tx * self.ty
        self.full = self.n // world  # tiles every rank has
        self.rem = self.n % world    # ranks below this have one more
        self.mine = tile_ids(width, height, rank, world).to(device)
        # pixel index (row-major over the surface) of every pixel of this rank's tiles, tile by tile: one gather
        # brings the tiles out of the surface without re-laying out the whole frame first
        t = self.mine.view(-1, 1, 1)
        rows = (t // self.tx) * TILE + torch.arange(TILE, device=device).view(1, TILE, 1)
        cols = (t % self.tx) * TILE + torch.arange(TILE, device=device).view(1, 1, TILE)
        self.pixels = (rows * width + cols).reshape(-1)
        if rank == 0:
            per = self.full + (1 if self.rem else 0)
            self.buf = torch.zeros((world, per, TILE, TILE, channels), dtype=dtype, device=device)
            self.frame = torch.zeros((self.ty, self.tx, TILE, TILE, channels), dtype=dtype, device=device)

    def tiles(self, surface):
        """[h, w, c] surface -> [tiles, 8, 8, c] of this rank's tiles (a copy)."""
        return surface.view(self.h * self.w, self.c).index_select(0, self.pixels).view(-1, TILE, TILE, self.c)

    def gather(self, surface, assemble=True):
        """Every rank passes its own surface (only its tiles are read).  Rank 0 returns the assembled [h, w, c] frame,
        or, with assemble=False, leaves the tiles as they arrived (rank by rank) for a later assembled() call."""
        if self.rank == 0:
            # rank 0's own tiles go straight into their place in the receive buffer
            own = self.buf[0, : self.pixels.numel() // (TILE * TILE)].view(-1, self.c)
            torch.index_select(surface.view(self.h * self.w, self.c), 0, self.pixels, out=own)
        else:
            mine = self.tiles(surface)
        # Both sides use the batched point-to-point call: on RCCL it puts rank 0's N-1 receives into ONE group, so
        # they run concurrently over the peers' separate xGMI links, and it keeps sender and receiver on the same
        # communicator.
        if self.rank != 0:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, 0)]):
                q.wait()
            return None
        ops = [dist.P2POp(dist.irecv, self.buf[r, :tile_count(self.w, self.h, r, self.world)], r) for r in range(1, self.world)]
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for q in reqs:
            q.wait()
        return self.assembled() if assemble else None

    def assembled(self):
        """Rank 0: the latest gathered frame as [h, w, c] (a copy).  The gather leaves the tiles rank by rank
        (tile k * world + r sits in buf[r, k]); here they are interleaved into tile order and laid out as rows."""
        flat = self.frame.view(self.n, TILE, TILE, self.c)
        flat[: self.full * self.world].view(self.full, self.world, TILE, TILE, self.c).copy_(self.buf[:, : self.full].transpose(0, 1))
        if self.rem:
            flat[self.full * self.world:] = self.buf[: self.rem, self.full]
        return self.frame.permute(0, 2, 1, 3, 4).reshape(self.h, self.w, self.c)
