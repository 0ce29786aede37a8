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


def interleaved_assignment(width, height, world):
    """Tile t on rank t % world: every rank covers the whole image thinly."""
    return [tile_ids(width, height, r, world) for r in range(world)]


def balanced_assignment(cost, world):
    """Per-rank tile lists of (nearly) equal COST and equal size, from a per-tile cost table (e.g. the box + triangle
    tests of the previous frame, rt_set_tile_cost_buffer).  Tiles sorted by falling cost are dealt out in rounds of
    `world`; within a round the rank that is behind gets the most expensive tile (the classic longest-processing-time
    rule restricted to equal counts, so that the gather buffers keep their size).  Every list is ordered expensive
    first, which is also the order the trace kernels should take the tiles in."""
    cost = torch.as_tensor(cost, dtype=torch.float64).flatten().cpu()
    n = cost.numel()
    order = torch.argsort(cost, descending=True, stable=True)
    lists = [[] for _ in range(world)]
    load = [0.0] * world
    for start in range(0, n, world):
        chunk = order[start:start + world].tolist()
        ranks = sorted(range(world), key=lambda r: (len(lists[r]), load[r], r))  # the short-handed first, then the least loaded
        for t, r in zip(chunk, ranks):
            lists[r].append(t)
            load[r] += float(cost[t])
    return [torch.tensor(l, dtype=torch.int64) for l in lists]


def quantize8(x):
    """float -> 8 bit with the reference's dump rule clamp((int)(v * 256), 0, 255) (cudaTracer/PicDumper.h:43-55); NaN -> 0."""
    return (x * 256.0).nan_to_num(0.0, 255.0, 0.0).clamp_(0.0, 255.0).to(torch.uint8)


def center_out(tiles, width, height):
    """The same tiles, in the order the library itself hands tiles out: by the distance of the tile's centre from the
    image centre (ties by tile number).  WHICH tiles a rank renders is what balances the load; the ORDER within its list
    decides how the rays of neighbouring tiles share the caches and whether the long rays start first - a list left in
    order of falling cost renders about a tenth slower."""
    t = torch.as_tensor(tiles, dtype=torch.int64).flatten()
    tx, _ = tile_grid(width, height)
    dx = (t % tx) * 2 * TILE + TILE - width
    dy = (t // tx) * 2 * TILE + TILE - height
    key = (dx * dx + dy * dy) * (int(t.max()) + 1 if t.numel() else 1) + t
    return t[torch.argsort(key)]


class TileGather:
    """Collects the tiles each rank rendered into the full frame on rank 0.

    `assignment`: per-rank lists of frame tile indices (default: interleaved, tile t on rank t % world); every tile of
    the frame must appear exactly once.  Frame sizes need not be multiples of 8: edge tiles travel whole, the frame is
    cropped when it is assembled.  `quantize=True` sends 8-bit pixels (the PNG dump's rule) instead of floats: a
    quarter of the bytes over xGMI for a consumer that only wants the picture.

    Rank 0 receives every peer's tiles straight into one [world, tiles per rank, 8, 8, c] buffer (one irecv per peer,
    all in flight together) and brings them into tile order with a single gather.

    `ctx` (an api.Context of this rank's GPU): float4 surfaces in device memory are then packed and, on rank 0,
    assembled by the library's own kernels (rt_tiles_pack / rt_tiles_scatter, enqueued on the context's stream, which
    must be torch's current stream) instead of torch index arithmetic: one launch per end and rank, nothing
    allocated per frame.  `device` stays the device the transfers run on (the GPU for RCCL; "cpu" for a rehearsal over
    gloo, where the packed tiles cross to the host and back)."""

    def __init__(self, width, height, rank, world, device, channels=4, dtype=torch.float32, assignment=None, quantize=False, ctx=None):
        self.w, self.h, self.rank, self.world, self.device = width, height, rank, world, device
        self.ctx = ctx
        if ctx is not None and (channels != 4 or dtype != torch.float32):
            raise ValueError("the library's tile kernels move float4 pixels")
        self.tx, self.ty = tile_grid(width, height)
        self.c = channels
        self.n = self.tx * self.ty
        self.quantize = quantize
        self.dtype = torch.uint8 if quantize else dtype
        self.buf = self.frame = None
        self.set_assignment(assignment if assignment is not None else interleaved_assignment(width, height, world))

    def set_assignment(self, assignment):
        """New per-rank tile lists (e.g. after re-balancing); every rank must call it with the same lists."""
        if len(assignment) != self.world:
            raise ValueError("one tile list per rank")
        device = self.device
        lists = [torch.as_tensor(a, dtype=torch.int64).flatten() for a in assignment]
        allt = torch.cat(lists)
        if allt.numel() != self.n or not torch.equal(torch.sort(allt).values, torch.arange(self.n)):
            raise ValueError("the tile lists must cover every tile of the frame exactly once")
        self.counts = [int(l.numel()) for l in lists]
        self.per = max(self.counts)
        self.mine = lists[self.rank].to(device)
        # pixel index (row-major over the surface) of every pixel of this rank's tiles, tile by tile: one gather
        # brings the tiles out of the surface without re-laying out the whole frame first.  Pixels of an edge tile
        # that lie outside the frame read the nearest pixel inside; they are cropped away on rank 0.
        t = self.mine.view(-1, 1, 1)
        rows = ((t // self.tx) * TILE + torch.arange(TILE, device=device).view(1, TILE, 1)).clamp_(max=self.h - 1)
        cols = ((t % self.tx) * TILE + torch.arange(TILE, device=device).view(1, 1, TILE)).clamp_(max=self.w - 1)
        self.pixels = (rows * self.w + cols).reshape(-1)
        if self.rank == 0:
            if self.buf is None or self.buf.shape[1] != self.per:
                self.buf = torch.zeros((self.world, self.per, TILE, TILE, self.c), dtype=self.dtype, device=device)
            if self.frame is None:
                self.frame = torch.zeros((self.ty, self.tx, TILE, TILE, self.c), dtype=self.dtype, device=device)
            # where tile t sits in the receive buffer (rank r, position k) as one flat index
            where = torch.empty(self.n, dtype=torch.int64)
            for r, l in enumerate(lists):
                where[l] = r * self.per + torch.arange(l.numel())
            self.where = where.to(device)
        if self.ctx is not None:
            gpu = torch.device("cuda", torch.cuda.current_device())
            self.k_tiles = lists[self.rank].to(torch.int32).to(gpu)
            self.k_packed = torch.empty((max(self.counts[self.rank], 1), TILE, TILE, self.c), dtype=self.dtype, device=gpu)
            if self.rank == 0:
                self.k_lists = [l.to(torch.int32).to(gpu) for l in lists]
                self.k_frame = torch.zeros((self.h, self.w, self.c), dtype=self.dtype, device=gpu)

    def tiles(self, surface):
        """[h, w, c] surface -> [tiles, 8, 8, c] of this rank's tiles (8-bit if the gather quantizes).  With the
        library's kernel the result is this object's own buffer, overwritten by the next call."""
        if self.ctx is not None and surface.is_cuda:
            n = self.counts[self.rank]
            self.ctx.tiles_pack(surface.data_ptr(), surface.stride(0) * surface.element_size(), self.w, self.h, self.k_tiles.data_ptr(), n,
                                self.k_packed.data_ptr(), self.quantize)
            return self.k_packed[:n]
        t = surface.view(self.h * self.w, self.c).index_select(0, self.pixels).view(-1, TILE, TILE, self.c)
        return quantize8(t) if self.quantize else t

    def gather(self, surface, assemble=True):
        """Every rank passes its own surface (only its tiles are read).  Rank 0 returns the assembled [h, w, c] frame,
        or, with assemble=False, leaves the tiles as they arrived (rank by rank) for a later assembled() call."""
        mine = self.tiles(surface)
        if mine.device != torch.device(self.device):
            mine = mine.to(self.device)  # packed on the GPU, sent from the host: a rehearsal over gloo
        # Both sides use the batched point-to-point call: on RCCL it puts rank 0's N-1 receives into ONE group, so
        # they run concurrently over the peers' separate xGMI links, and it keeps sender and receiver on the same
        # communicator.
        if self.rank != 0:
            if mine.numel():
                for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), 0)]):
                    q.wait()
            return None
        self.buf[0, : self.counts[0]].copy_(mine)
        ops = [dist.P2POp(dist.irecv, self.buf[r, : self.counts[r]], r) for r in range(1, self.world) if self.counts[r]]
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for q in reqs:
            q.wait()
        return self.assembled() if assemble else None

    def assembled(self):
        """Rank 0: the latest gathered frame as [h, w, c]: tiles into tile order, rows laid out, edge cropped.  (With the
        library's kernels: one scatter launch per rank into this object's frame tensor, which the next call rewrites.)"""
        if self.ctx is not None:
            buf = self.buf if self.buf.is_cuda else self.buf.to(self.k_frame.device)
            pitch = self.w * self.c * self.k_frame.element_size()
            for r in range(self.world):
                if self.counts[r]:
                    self.ctx.tiles_scatter(buf[r].data_ptr(), self.k_lists[r].data_ptr(), self.counts[r], self.k_frame.data_ptr(), pitch, self.w, self.h,
                                           self.quantize)
            return self.k_frame
        flat = self.frame.view(self.n, TILE, TILE, self.c)
        torch.index_select(self.buf.view(self.world * self.per, TILE, TILE, self.c), 0, self.where, out=flat)
        full = self.frame.permute(0, 2, 1, 3, 4).reshape(self.ty * TILE, self.tx * TILE, self.c)
        return full[: self.h, : self.w].contiguous()


class FramePipeline:
    """Frames in flight together with their gathers.  Frame i renders into surface i % F (a swap chain of F
    surfaces); as soon as F frames are in flight the oldest one is gathered while the newer ones keep rendering, and
    drain() gathers the ones still in flight.

    render(surface, i)   enqueues frame i (every rank: its own tiles)
    wait_frame(ago)      orders the gather's stream behind the frame issued `ago` calls ago (no-op on the CPU)
    gatherer             a TileGather, or None on a single rank
    to_comm(surface)     surface as the gather's backend wants it (identity on RCCL, .cpu() for a gloo rehearsal)
    on_gathered(i)       called on every rank after frame i has been gathered (tests look at the frame there)"""

    def __init__(self, surfaces, render, wait_frame=None, gatherer=None, to_comm=None, on_gathered=None):
        self.surfaces, self.nf = surfaces, len(surfaces)
        self.render, self.wait_frame, self.gatherer = render, wait_frame, gatherer
        self.to_comm, self.on_gathered = to_comm, on_gathered
        self.issued = 0

    def _collect(self, frames_ago):
        idx = self.issued - 1 - frames_ago
        if self.wait_frame:
            self.wait_frame(frames_ago)
        s = self.surfaces[idx % self.nf]
        self.gatherer.gather(self.to_comm(s) if self.to_comm else s, assemble=False)
        if self.on_gathered:
            self.on_gathered(idx)

    def step(self):
        self.render(self.surfaces[self.issued % self.nf], self.issued)
        self.issued += 1
        if self.gatherer is not None and self.issued >= self.nf:
            self._collect(self.nf - 1)  # the oldest frame in flight; the newer ones keep running meanwhile

    def drain(self):
        if self.gatherer is not None:
            for ago in range(min(self.nf - 1, self.issued) - 1, -1, -1):
                self._collect(ago)
        self.issued = 0
