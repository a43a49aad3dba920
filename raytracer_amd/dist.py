"""Image-tile data parallelism over the GPUs of one node (one process per GPU, torch.distributed).

The frame is cut into 8x8-pixel tiles dealt round-robin to the ranks (cost per pixel varies by two
orders of magnitude across an image, so fine interleaving is the load balancer).  Every rank holds the
whole scene and traces only its tiles; there is exactly one exchange per frame (scenes with planar lights have
one more, in the middle: the per-pixel hit counts that fix the order of the reference's rand() draws):

  1. all-reduce of the per-channel colour range (MAX over 6 floats: the negated minima and the maxima): the tone map divides by
     the GLOBAL range (src/raytrace.c:396-399, 433-435 of the reference), so it cannot start earlier;
  2. every rank tone-maps its own tiles to RGBA8;
  3. the RGBA8 tiles (4 B/pixel) are gathered to rank 0, which de-interleaves them into the image.

On ROCm the "nccl" backend is RCCL; the gather is a set of direct peer->root transfers over xGMI.
The same functions run on CPU tensors with the gloo backend (used by the CPU tests).

A sequence of frames can be kept several deep (FramePipeline): every frame in flight has its own context, queues,
stream and output buffers, so one frame's exchange and the latency-bound ends of its walk kernels run under the next
frame's kernels.  One rank's share of a 4K frame is a few tickets per resident wavefront (2.6 on eight GPUs), and a
walk kernel there lasts as long as its slowest ticket while most of the machine idles: two frames in flight give
1.35 -> 1.05 ms per frame for one rank of eight, 2.10 -> 1.75 of four, 3.19 -> 2.95 of two (tools/frames_in_flight.py).
"""
import numpy as np

import torch
import torch.distributed as dist

TILE = 8
TILE_PIXELS = 64
# Tests set this to run the collectives even in a group of one rank (RCCL then sees exactly the tensors it gets with more).
FORCE_COLLECTIVES = False


def tiles_of(width, height):
    return ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)


def tile_count(ntiles, rank, world):
    """Number of tiles rank owns when tiles rank, rank+world, ... belong to it."""
    return (ntiles - rank + world - 1) // world if rank < ntiles else 0


_sign_cache = {}


def _signs(like):
    key = (like.device, like.dtype)
    if key not in _sign_cache:
        _sign_cache[key] = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=like.dtype, device=like.device)
    return _sign_cache[key]


def exchange_minmax(minmax, group=None):
    """minmax: tensor [6] = min rgb, max rgb of this rank.  Returns the global range on every rank."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        return minmax
    # one collective: MAX of (-min, max); negation is exact, so this is the MIN of the minima
    sign = _signs(minmax)
    both = minmax * sign
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    return both * sign


def all_gather_counts(local, ntiles, group=None):
    """Planar lights: every rank's per-pixel shaded-hit counts (tile-major, tile_count(rank)*64 each) on every rank.
    Returns (tensor [world * per_rank], per_rank, total hits of the frame)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    per_rank = tile_count(ntiles, 0, world) * TILE_PIXELS
    send = local
    if local.numel() < per_rank:
        send = torch.zeros(per_rank, dtype=local.dtype, device=local.device)
        send[:local.numel()] = local
    if world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
        return send, per_rank, int(send.sum().item())
    out = torch.empty(per_rank * world, dtype=local.dtype, device=local.device)
    dist.all_gather(list(out.split(per_rank)), send, group=group)
    return out, per_rank, int(out.sum().item())


def gather_tiles(local, ntiles, elems_per_pixel=1, dst=0, group=None):
    """Gathers tile-major buffers (tile_count(rank)*64*elems elements each) to `dst`.

    Returns on dst a tensor laid out rank-major (rank 0's tiles, then rank 1's, ...) with exactly
    tile_count(r)*64*elems elements per rank, which is the layout rtgpu_untile reads; None elsewhere.
    Ranks own different tile counts when ntiles % world != 0, so buffers are padded to the largest.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
        return local
    per_tile = TILE_PIXELS * elems_per_pixel
    most = tile_count(ntiles, 0, world) * per_tile
    send = local
    if local.numel() < most:
        send = torch.zeros(most, dtype=local.dtype, device=local.device)
        send[:local.numel()] = local
    parts = whole = None
    if rank == dst:  # the parts are views of one buffer: nothing to concatenate when every rank owns as many tiles
        whole = torch.empty(most * world, dtype=local.dtype, device=local.device)
        parts = list(whole.split(most))
    dist.gather(send, parts, dst=dst, group=group)
    if rank != dst:
        return None
    if ntiles % world == 0:
        return whole
    return torch.cat([parts[r][:tile_count(ntiles, r, world) * per_tile] for r in range(world)])


def split_tiles_reference(image, rank, world):
    """numpy: the tile-major buffer rank would produce for a row-major [h, w, ...] image."""
    h, w = image.shape[:2]
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    pad = np.zeros((ty * TILE, tx * TILE) + image.shape[2:], image.dtype)
    pad[:h, :w] = image
    tiles = pad.reshape((ty, TILE, tx, TILE) + image.shape[2:]).swapaxes(1, 2).reshape((ty * tx, TILE_PIXELS) + image.shape[2:])
    return tiles[rank::world].copy()


def untile_reference(gathered, world, width, height):
    """numpy inverse of the rank-major tile layout: gathered [ntiles*64, ...] -> [h, w, ...]."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    ntiles = tx * ty
    g = gathered.reshape((ntiles, TILE_PIXELS) + gathered.shape[1:])
    tiles = np.zeros_like(g)
    at = 0
    for r in range(world):
        n = tile_count(ntiles, r, world)
        tiles[r::world] = g[at:at + n]
        at += n
    img = tiles.reshape((ty, tx, TILE, TILE) + gathered.shape[1:]).swapaxes(1, 2).reshape((ty * TILE, tx * TILE) + gathered.shape[1:])
    return img[:height, :width].copy()


class FrameRenderer:
    """Per-rank driver of one frame: trace own tiles, exchange the range, tone-map, gather, untile."""

    def __init__(self, renderer, scene, rank=0, world=1, group=None):
        self.r, self.scene, self.rank, self.world, self.group = renderer, scene, rank, world, group
        dev = torch.device("cuda", renderer.device)
        self.dev = dev
        renderer.upload(scene)
        renderer.frame_begin()
        self.w, self.h = scene.width, scene.height
        self.ntiles = tiles_of(self.w, self.h)
        n = renderer.tile_count(rank, world) * TILE_PIXELS
        self.npix_local = n
        self.rgb = torch.zeros(max(n, 1) * 3, dtype=torch.float32, device=dev)
        self.ids = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.minmax = torch.zeros(6, dtype=torch.float32, device=dev)
        self.rgba = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.counts = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.image = torch.zeros(self.w * self.h, dtype=torch.int32, device=dev) if rank == 0 else None
        renderer.set_stream(torch.cuda.current_stream(dev))

    def frame(self):
        """One frame: everything rtVisualizedSceneRaytrace + rtVisualizedSceneToBitmap do, on `world` GPUs.
        Returns the row-major RGBA8 image (int32 tensor, w*h) on rank 0."""
        r = self.r
        r.frame_begin()                                    # preprocess + grid build (replicated)
        if self.scene.npl > 0 and (self.world > 1 or FORCE_COLLECTIVES):
            # planar lights draw rand() in pixel order: the ranks exchange their hit counts between the two halves
            r.render_trace(self.rank, self.world)
            r.hit_counts(self.counts)
            every, per_rank, total = all_gather_counts(self.counts[:self.npix_local], self.ntiles, self.group)
            r.render_shade(self.rgb, self.ids, self.minmax, every, per_rank, total)
        else:
            r.render(self.rgb, self.ids, self.minmax, self.rank, self.world)
        mm = exchange_minmax(self.minmax, self.group)      # exchange 1: 2 x 3 floats
        r.tonemap(self.rgb, self.npix_local, mm, self.scene.gamma, self.rgba)
        self.global_minmax = mm
        gathered = gather_tiles(self.rgba[:self.npix_local], self.ntiles, 1, 0, self.group)  # exchange 2: 4 B/pixel
        if self.rank == 0:
            r.untile(gathered, self.world, 4, self.image)
        return self.image


def default_depth(world):
    """Frames in flight on each of `world` GPUs for a workload of many launches per frame (secondary rays; bench.py's
    config5 legs).  A share of a frame leaves most of the machine idle for most of a walk kernel: two in flight (one rank
    of eight 1.35 -> 1.05 ms per frame; three: 1.14).  A whole frame of that kind fills one GPU by itself."""
    return 1 if world <= 1 else 2


def headline_depth(world):
    """Frames in flight on each of `world` GPUs for a frame of few launches (no secondary rays: bench.py's headline
    workload).  The ends of its walk kernels are latency-bound, so even a whole frame gains from a second one beside it
    (temple 4K, one GPU, one / two / three in flight: 2.14 / 1.80 / 1.82 ms per frame); an eighth of a frame is short
    against those ends and against the exchange, and three hide more of them (one rank of eight, two / three in flight:
    0.57 / 0.47 ms; of four: 0.72 / 0.77; of two: 1.19 / 1.19; with a hardware queue for every stream, GPU_MAX_HW_QUEUES=16)."""
    return 3 if world >= 8 else 2


class FramePipeline:
    """A sequence of frames, `depth` of them in flight on this rank's GPU.  Frame k runs in slot k % depth: a context of
    its own (scene, grid buffers, queues), a stream of its own and its own output buffers.  Every rank issues the
    frames, and with them the collectives, in the same order; the collectives of one frame wait for that frame's kernels
    only, so the next frame's kernels run beside them.

    make_renderer() returns a fresh Renderer on this rank's device for every slot.

    Streams and hardware queues: every slot keeps two streams busy (its own and the one its context sets the next frame up
    on).  HIP shares a few hardware queues among a process's streams (GPU_MAX_HW_QUEUES, four by default) and streams on
    one queue run one after the other, so a process that keeps frames in flight should start with GPU_MAX_HW_QUEUES=16
    in its environment (before its first HIP call; bench.py does: temple 4K two in flight 1.96 -> 1.80 ms per frame)."""

    def __init__(self, make_renderer, scene, rank=0, world=1, depth=None, group=None):
        depth = default_depth(world) if depth is None else int(depth)
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.depth, self.issued = depth, 0
        self.renderers, self.slots, self.streams = [], [], []
        for _ in range(depth):
            r = make_renderer()
            r.set_option("light_lists_after", 0)  # (a sequence of frames: the scene's light lists from its first frame on)
            if depth > 1:
                # frames in flight share the GPU: the library then sizes the walk kernels of scenes without secondary rays so
                # that the other frame's kernels and set-up find room beside them (rtgpu.hip: waves_of)
                r.set_option("contexts_on_gpu", depth)
            dev = torch.device("cuda", r.device)
            s = torch.cuda.current_stream(dev) if depth == 1 else torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                self.slots.append(FrameRenderer(r, scene, rank, world, group))
            self.renderers.append(r)
            self.streams.append(s)
        self.dev = dev

    def frame(self):
        """Enqueues the next frame.  Returns (image, stream): the row-major RGBA8 image of rank 0 (None elsewhere), complete
        once everything enqueued on `stream` so far has run.  The slot's previous image (depth frames ago) is overwritten."""
        i = self.issued % self.depth
        self.issued += 1
        with torch.cuda.stream(self.streams[i]):
            img = self.slots[i].frame()
        return img, self.streams[i]

    def finish(self):
        """Makes the current stream wait for every frame issued so far."""
        cur = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            if s != cur:
                cur.wait_stream(s)

    def close(self):
        for r in self.renderers:
            r.close()
