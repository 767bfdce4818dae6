"""Multi-GPU host logic: one process per GPU, torch.distributed (RCCL) for
the two exchange steps the path really has —
  * BVH replication: built once on rank 0, the flat buffers travel as one blob, scatter + all-gather over xGMI;
  * framebuffer gather: each rank renders its 16-row stripes (stripe s -> rank s % world) into
    a PACKED buffer; the equal-sized packed buffers are gathered — to rank 0 only (exchange="root": the viewer's rank,
    grouped send/recv, every byte crosses one xGMI link once, 1/world of an all-gather's traffic) or to every rank
    (exchange="all": all_gather) — and de-interleaved by one kernel wherever they arrive.
Pixels are independent: there is no collective inside the traversal itself.

torch is plumbing only (device buffers + collectives); compute is behind the C ABI.  The
`backend` object abstracts "render my packed stripes" / "de-interleave" so that the
world_size-2 gloo tests can drive the same host logic on CPU with the oracle as backend.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import packed_rows, render_opts

STRIPE_ROWS = 16


def rows_per_rank(height, world, stripe_rows=STRIPE_ROWS):
    """Packed-buffer height every rank allocates: rank 0 always owns the most rows."""
    return packed_rows(height, stripe_rows, 0, world)


def stripe_rows_of(height, rank, world, stripe_rows=STRIPE_ROWS):
    """Global row numbers owned by `rank`, in packed order."""
    ys = [y for y in range(height) if (y // stripe_rows) % world == rank]
    return np.asarray(ys, dtype=np.int64)


class GpuBackend:
    """Renders through librtcore.so into torch-owned device buffers (no host copies)."""

    def __init__(self, ctx, device, frames_in_flight=2):
        self.ctx, self.device, self.fif = ctx, device, max(1, int(frames_in_flight))
        ctx.set_frames_in_flight(self.fif)     # the tail of frame i overlaps the head of frame i+1 (rtcore.h)
        # the context's own HIP stream, seen from torch: the collective and the de-interleave of a frame are queued on it
        self.ext = torch.cuda.ExternalStream(ctx.stream(), device=device)
        self._side = None

    def alloc(self, rows, width):
        return torch.zeros((rows, width), dtype=torch.int32, device=self.device)

    def render_packed(self, U, opts, packed, want_stats=True):
        """want_stats=True synchronises the context stream (stats are read back); False is fully asynchronous."""
        return self.ctx.render_device(U, opts, d_rgba8=packed.data_ptr(), want_stats=want_stats)

    def gather(self, gathered, packed, keep_latest, root_only=False):
        """Gather of a finished frame's stripes on the context stream: the stream first waits for that frame
        (all asynchronous frames issued so far, except the keep_latest most recent ones), then torch queues the
        collective behind it (on this stream, or on its own communication stream ordered against it).
        root_only: `gathered` is None on every rank but 0."""
        self.ctx.frames_join(int(keep_latest))
        with torch.cuda.stream(self.ext):
            gather_stripes(gathered, packed, root_only)

    def deinterleave(self, gathered, image, width, height, world, rpr):
        self.ctx.deinterleave_device(gathered.data_ptr(), image.data_ptr(), width, height, 4, STRIPE_ROWS, world, rpr)

    def gather_window(self, gathered, staging, packed, keep_latest, root_only, world, window):
        """gather() of the frame's WINDOW only (exchange_window): a strided copy into `staging`, then the collective, on the
        context stream behind the frame.  -> the gathered windows [world, n_rows, n_cols] (None off the root, or if empty)."""
        col0, n_cols, group0, n_rows = window
        self.ctx.frames_join(int(keep_latest))
        if n_cols == 0 or n_rows == 0:
            return None
        with torch.cuda.stream(self.ext):
            sub = staging.view(-1)[:n_rows * n_cols].view(n_rows, n_cols)
            r0 = group0 * STRIPE_ROWS
            sub.copy_(packed[r0:r0 + n_rows, col0:col0 + n_cols])
            g = gathered.view(-1)[:world * n_rows * n_cols].view(world * n_rows, n_cols) if gathered is not None else None
            gather_stripes(g, sub, root_only)
        return g

    def assemble(self, g, image, width, height, world, window, fill):
        col0, n_cols, group0, n_rows = window
        self.ctx.assemble_device(g.data_ptr() if g is not None else 0, image.data_ptr(), width, height, STRIPE_ROWS, world,
                                 col0 if g is not None else 0, n_cols if g is not None else 0, group0, n_rows if g is not None else 0,
                                 n_rows * n_cols, fill)

    def finish(self):
        self.ctx.synchronize()
        if self._side is not None:
            self._side.synchronize()
        torch.cuda.current_stream(self.device).synchronize()

    # ---- batches (BatchedFrames): B frames per pipeline, their exchange on a side stream
    def render_packed_batch(self, Us, opts, packed):
        """packed: [B, rows, width]; frame f of the batch goes to packed[f] (rt_render_batch_device, asynchronous)."""
        base, step = packed.data_ptr(), packed.stride(0) * packed.element_size()      # (one tensor call, not one per frame)
        self.ctx.render_batch_device(Us, opts, [base + f * step for f in range(len(Us))])

    def side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def mark(self):
        """An event behind everything queued on the context stream so far."""
        ev = torch.cuda.Event()
        ev.record(self.ext)
        return ev

    def mark_batch(self, keep_latest):
        """An event behind the batch issued `keep_latest` batches ago.  One pipeline in flight: batches run on the context
        stream itself, so the event is recorded right after the batch was queued (keep_latest is 0 then).  Several in flight
        (rt_set_frames_in_flight): the context stream first waits for all batches but the keep_latest newest."""
        if self.fif > 1:
            self.ctx.frames_join(int(keep_latest))
        return self.mark()

    def wait_for(self, ev):
        """The context stream waits for `ev` (device-side; the host does not block)."""
        if ev is not None:
            self.ext.wait_event(ev)

    def frame_rect(self, U):
        return self.ctx.frame_rect(U)

    def exchange_batch_window(self, gathered, staging, packed, images, after, root_only, width, height, world, window, fill):
        """exchange_batch with WINDOWS: only columns / rows `window` (exchange_window) of every rank's packed stripes cross the
        links - a strided copy into `staging`, ONE collective of [B, n_rows, n_cols] per rank, and rt_assemble_device_on
        builds each full image from the gathered windows and `fill` (the value of a missed pixel) everywhere else.
        gathered / staging: flat buffers at least as large as the windows (gathered: world x); -> the event that ends it."""
        col0, n_cols, group0, n_rows = window
        n = packed.shape[0]
        side = self.side_stream()
        side.wait_event(after)
        with torch.cuda.stream(side):
            g = None
            if n_cols > 0 and n_rows > 0:
                per_rank = n * n_rows * n_cols
                sub = staging.view(-1)[:per_rank].view(n, n_rows, n_cols)
                r0 = group0 * STRIPE_ROWS
                sub.copy_(packed[:, r0:r0 + n_rows, col0:col0 + n_cols])
                if gathered is not None:
                    g = gathered.view(-1)[:world * per_rank].view(world * n, n_rows, n_cols)
                gather_stripes(g, sub, root_only)
            if images is not None:
                for f in range(n):    # frame f of rank r: window (r * n + f) of `g`
                    self.ctx.assemble_device(g.data_ptr() + 4 * f * n_rows * n_cols if g is not None else 0, images[f].data_ptr(), width, height,
                                             STRIPE_ROWS, world, col0, n_cols, group0, n_rows, n * n_rows * n_cols, fill, stream=side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def exchange_batch(self, gathered, packed, images, after, root_only, width, height, world, rpr):
        """Gather of a whole batch ([B, rpr, width] per rank, ONE collective) and the de-interleave of its B frames, all on
        the side stream behind event `after`: the context stream is free to traverse the next batch meanwhile.
        -> the event that ends the exchange (the packed / gathered buffers may be reused behind it)."""
        side = self.side_stream()
        side.wait_event(after)
        with torch.cuda.stream(side):
            gather_stripes(gathered, packed, root_only)
            if images is not None:
                n = packed.shape[0]
                for f in range(n):    # frame f of rank r sits at row (r * n + f) * rpr of `gathered`
                    self.ctx.deinterleave_device(gathered.data_ptr() + 4 * f * rpr * width, images[f].data_ptr(), width, height, 4,
                                                 STRIPE_ROWS, world, n * rpr, stream=side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev


def _through_host(t):
    """gloo groups (rehearsals of the N > 1 path with several ranks on ONE GPU, where RCCL refuses to run) move device tensors
    through host memory: torch's gloo backend accepts device tensors but hung now and then on an idle device here (all ranks
    inside the same all_gather).  RCCL groups never take this path."""
    return t is not None and t.is_cuda and dist.get_backend() == "gloo"


def all_gather_into(out, chunk):
    if _through_host(chunk):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, chunk.cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, chunk)


def exchange_window(rects, width, height, world, rpr, stripe_rows=STRIPE_ROWS):
    """The part of every rank's packed stripes that a batch has to exchange: -> (col0, n_cols, group0, n_rows).
    rects: the frames' (x0, y0, x1, y1) screen rectangles (inclusive) outside of which every pixel is a miss (rt_frame_rect).
    Columns [col0, col0 + n_cols) (16-byte aligned) of the packed rows [group0 * stripe_rows, group0 * stripe_rows + n_rows)
    hold the union of the rectangles on every rank: stripe group g is the image rows [g * world * stripe_rows, (g + 1) * world *
    stripe_rows), one stripe of every rank, so whole groups keep the window the same on all ranks.  n_cols = 0: nothing to send."""
    live = [(max(r[0], 0), max(r[1], 0), min(r[2], width - 1), min(r[3], height - 1)) for r in rects]      # (clipped to the image)
    live = [r for r in live if r[2] >= r[0] and r[3] >= r[1]]
    if not live:
        return 0, 0, 0, 0
    x0, y0 = min(r[0] for r in live), min(r[1] for r in live)
    x1, y1 = max(r[2] for r in live), max(r[3] for r in live)
    col0 = x0 & ~3
    n_cols = min(width - col0, (x1 + 1 - col0 + 3) & ~3)
    rows_per_group = stripe_rows * world
    group0, group1 = y0 // rows_per_group, y1 // rows_per_group + 1
    n_rows = min(group1 * stripe_rows, rpr) - group0 * stripe_rows
    return col0, n_cols, group0, n_rows


def gather_stripes(gathered, packed, root_only):
    """The frame's one exchange step.  gathered: [world * rows_per_rank, width] (rank-major), or None off the root."""
    if not root_only:
        all_gather_into(gathered, packed)
    elif _through_host(packed):
        world = dist.get_world_size()
        h = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype) if dist.get_rank() == 0 else None
        dist.gather(packed.cpu(), list(h.unbind(0)) if h is not None else None, dst=0)
        if h is not None:
            gathered.view(h.shape).copy_(h)
    elif dist.get_rank() == 0:
        dist.gather(packed, list(gathered.view(dist.get_world_size(), *packed.shape).unbind(0)), dst=0)
    else:
        dist.gather(packed, None, dst=0)


def replicate_bvh(ctx, rank, world, device):
    """Rank 0's flat BVH (nodes, triPos, triNor, and the BVH-order surface records when the scene has them) to every rank:
    ONE blob, scattered in `world` equal chunks (rank 0 sends 1/world of it down each of its links at once) and then
    all-gathered (every link carries chunks in parallel) - instead of rooted broadcasts that push the whole blob (16 GB
    for cfg 5) through rank 0's links once per buffer.  Every other rank adopts the blob's segments in place.
    Returns the tensors that back the adopted BVH (keep them alive).
    Materials and textures are small host-side descriptions: every rank sets them itself."""
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == 0:
        T, N = ctx.counts()
        uv_ptr, uv_bytes = ctx.tri_uv_device() if hasattr(ctx, "tri_uv_device") else (0, 0)
        meta[0], meta[1], meta[2], meta[3] = T, N, ctx.node_encoding(), 1 if uv_bytes else 0
    if _through_host(meta):
        h = meta.cpu()
        dist.broadcast(h, 0)
        meta.copy_(h)
    else:
        dist.broadcast(meta, 0)
    T, N, enc, has_uv = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
    seg = [32 * N, 48 * T, 48 * T] + ([32 * T] if has_uv else [])      # bytes; every offset is a multiple of 16
    off = [0]
    for n in seg:
        off.append(off[-1] + n)
    align = 256 * world
    total = (off[-1] + align - 1) // align * align
    blob = torch.empty(total, dtype=torch.uint8, device=device)
    base = blob.data_ptr()
    if rank == 0:
        ctx.export_bvh_device(base + off[0], base + off[1], base + off[2], N, T)
        if has_uv:
            ctx.export_tri_uv_device(base + off[3], T)
        if torch.device(device).type == "cuda":
            ctx.synchronize()           # the exports ran on the context stream; the collectives below run on torch's
    if world > 1:
        chunk = torch.empty(total // world, dtype=torch.uint8, device=device)
        if _through_host(chunk):
            h = torch.empty(total // world, dtype=torch.uint8)
            dist.scatter(h, list(blob.cpu().view(world, -1).unbind(0)) if rank == 0 else None, src=0)
            chunk.copy_(h)
        else:
            dist.scatter(chunk, list(blob.view(world, -1).unbind(0)) if rank == 0 else None, src=0)
        all_gather_into(blob, chunk)                    # (rank 0 receives what it already holds)
        del chunk
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if rank != 0:
        ctx.set_node_encoding(enc)          # v1 / v2 node records travel as the same 32-byte slots
        ctx.adopt_bvh_device(base + off[0], base + off[1], base + off[2], N, T)
        if has_uv:
            ctx.adopt_tri_uv_device(base + off[3], T)   # copied
    return [blob] if rank != 0 else []      # rank 0 renders from its own build: its copy of the blob is dropped here


class ShardedFrame:
    """One frame sharded as image-space stripes over `world` ranks and gathered on every rank.

    Asynchronous frame loops (step(want_stats=False)) are software-pipelined: step i queues the render of frame i
    first and only then the all-gather + de-interleave of frame i-1, so with two frames in flight (GpuBackend) the
    xGMI exchange and the tail of frame i-1 overlap the traversal of frame i.  Two packed / gathered buffer pairs
    alternate; frame i+2 reuses pair i only after all-gather(i) (a frame starts behind everything queued on the
    context stream before it)."""

    DEPTH = 2      # ring of packed / gathered buffers and frames kept un-gathered (more with GpuBackend(frames_in_flight > 2))

    def __init__(self, backend, width, height, rank, world, base_opts=None, force_collective=False, exchange="all", window_exchange=True):
        """force_collective: run the gather + de-interleave leg even at world == 1 (tests the stream plumbing on one GPU).
        exchange: "all" = every rank ends up with the frame (all_gather); "root" = only rank 0 does (self.image is None elsewhere)."""
        if exchange not in ("all", "root"):
            raise ValueError("exchange must be 'all' or 'root'")
        self.b, self.w, self.h, self.rank, self.world = backend, width, height, rank, world
        self.root_only = exchange == "root"
        self.has_frame = rank == 0 or not self.root_only
        self.collective = world > 1 or force_collective
        self.rpr = rows_per_rank(height, world) if world > 1 else height
        o = base_opts or render_opts()
        self.opts = render_opts(mode=o.mode, gi_samples=o.gi_samples, gi_bounces=o.gi_bounces, soft_samples=o.soft_samples,
                                max_stack=o.max_stack, layout=o.layout, flags=o.flags,
                                stripe_rows=STRIPE_ROWS if world > 1 else 0, stripe_rank=rank, stripe_world=world)
        # one output buffer per frame that can be in flight (also without a collective), at least two
        depth = max(self.DEPTH, getattr(backend, "fif", self.DEPTH))
        self.packed_ring = [backend.alloc(self.rpr, width) for _ in range(depth)]
        self.gathered_ring = ([backend.alloc(self.rpr * world, width) if self.has_frame else None for _ in range(depth)]
                              if self.collective else None)
        self.packed = self.packed_ring[0]
        self.image = (backend.alloc(height, width) if self.has_frame else None) if self.collective else self.packed
        self.n_steps = 0
        self.pending = []              # ring slots of the frames rendered (or rendering) but not yet gathered, oldest first
        # only the frame's screen rectangle is exchanged (see BatchedFrames); the compute.wgsl modes (rt_frame_rect)
        self.windowed = (bool(window_exchange) and self.collective and hasattr(backend, "gather_window") and o.mode <= 2)
        self.staging_ring = [backend.alloc(self.rpr, width) for _ in range(depth)] if self.windowed else None
        self.window = [None] * depth

    def _exchange(self, slot, keep_latest):
        packed, gathered = self.packed_ring[slot], self.gathered_ring[slot]
        if self.windowed:
            window, fill = self.window[slot]
            g = self.b.gather_window(gathered, self.staging_ring[slot], packed, keep_latest, self.root_only, self.world, window)
            if self.has_frame:
                self.b.assemble(g, self.image, self.w, self.h, self.world, window, fill)
            return
        if hasattr(self.b, "gather"):
            self.b.gather(gathered, packed, keep_latest, self.root_only)
        else:
            gather_stripes(gathered, packed, self.root_only)
        if self.has_frame:
            self.b.deinterleave(gathered, self.image, self.w, self.h, self.world, self.rpr)

    def step(self, U, want_stats=True):
        """Render my stripes, gather all stripes, de-interleave.  Returns this rank's render stats (or None).
        With want_stats=False nothing synchronises with the host and this frame's exchange is queued by the NEXT
        step (or finish()): call finish() before reading self.image."""
        slot = self.n_steps % len(self.packed_ring)
        self.n_steps += 1
        packed = self.packed_ring[slot]
        if self.windowed:
            rect, fill = self.b.frame_rect(U)
            self.window[slot] = (exchange_window([rect], self.w, self.h, self.world, self.rpr), fill)
        st = self.b.render_packed(U, self.opts, packed, want_stats) if not want_stats else self.b.render_packed(U, self.opts, packed)
        if not self.collective:
            self.image = packed                            # world == 1: the packed rows are the frame
            return st
        self.pending.append(slot)
        while len(self.pending) >= len(self.packed_ring):  # the oldest frame, while the newer ones are traversing
            self._exchange(self.pending.pop(0), keep_latest=len(self.pending))
        if want_stats:
            self.finish_exchange()
        return st

    def finish_exchange(self):
        while self.pending:
            self._exchange(self.pending.pop(0), keep_latest=len(self.pending))

    def finish(self):
        if self.collective:
            self.finish_exchange()
        if hasattr(self.b, "finish"):
            self.b.finish()


def batch_sizes(n, batch):
    """n frames as ceil(n / batch) pipelines of (nearly) equal size: 20 frames, 16 per pipeline at most -> [10, 10]."""
    if n <= 0:
        return []
    k = -(-n // batch)
    q, r = divmod(n, k)
    return [q + 1] * r + [q] * (k - r)


class BatchedFrames:
    """B frames of one image shape per step: this rank's stripes of all B frames are rendered by ONE pipeline
    (rt_render_batch_device: the launches' fixed cost and the idle tail of a small stripe share are paid once per batch,
    not once per frame), exchanged by ONE collective of [B, rows_per_rank, width] per rank and de-interleaved into B images.
    Software-pipelined like ShardedFrame: step i queues the traversal of batch i on the context stream, then the exchange
    of batch i-1 on a side stream, so xGMI transfer + de-interleave overlap the next batch's traversal; two packed /
    gathered buffer sets alternate, and batch i+2 starts behind the exchange that read set i.  With
    GpuBackend(frames_in_flight=2) the batches themselves rotate over two frame streams (the cull / compaction passes and the
    head of batch i+1 run beside the tail of batch i: 6-13 % more frames per second, tools/batch_gain.py).
    Backends without streams (the CPU oracle backend of the gloo tests) run the same steps synchronously."""

    def __init__(self, backend, width, height, rank, world, batch, base_opts=None, force_collective=False, exchange="root",
                 window_exchange=True):
        if exchange not in ("all", "root"):
            raise ValueError("exchange must be 'all' or 'root'")
        if not 1 <= batch <= 32:
            raise ValueError("batch must be in [1, 32] (RT_MAX_BATCH)")
        self.b, self.w, self.h, self.rank, self.world, self.batch = backend, width, height, rank, world, batch
        self.root_only = exchange == "root"
        self.has_frame = rank == 0 or not self.root_only
        self.collective = world > 1 or force_collective
        self.rpr = rows_per_rank(height, world) if world > 1 else height
        o = base_opts or render_opts()
        self.opts = render_opts(mode=o.mode, gi_samples=o.gi_samples, gi_bounces=o.gi_bounces, soft_samples=o.soft_samples,
                                max_stack=o.max_stack, layout=o.layout, flags=o.flags,
                                stripe_rows=STRIPE_ROWS if world > 1 else 0, stripe_rank=rank, stripe_world=world)
        alloc3 = lambda n, rows: backend.alloc(n * rows, width).view(n, rows, width)
        self.packed_ring = [alloc3(batch, self.rpr) for _ in range(2)]
        self.gathered_ring = ([alloc3(world * batch, self.rpr) if self.has_frame else None for _ in range(2)] if self.collective else None)
        self.images = (alloc3(batch, height) if self.has_frame else None) if self.collective else self.packed_ring[0]
        self.rendered = [None, None]   # per buffer set: event behind the batch's traversal
        self.free = [None, None]       # per buffer set: event behind the exchange that read it
        self.count = [0, 0]            # per buffer set: frames of the batch it holds
        self.prev = None
        self.n_steps = 0
        self.streams = hasattr(backend, "exchange_batch")
        # Only the union of the frames' screen rectangles is exchanged (the rest of every frame is background, known without
        # sending it): cfg 3's camera, 9 MB instead of 33 MB per frame.  Every rank derives the same window from the uniforms.
        self.windowed = bool(window_exchange) and self.collective and hasattr(backend, "exchange_batch_window") and o.mode <= 2   # (the fragment modes' rectangle is mirrored: whole stripes, as ShardedFrame)
        self.staging_ring = [backend.alloc(batch * self.rpr, width) for _ in range(2)] if self.windowed else None
        self.window = [None, None]     # per buffer set: ((col0, n_cols, group0, n_rows), fill)

    def step(self, Us):
        """Us: [batch, 44] uniforms.  Asynchronous; finish() before reading self.images (the frames of the LAST batch)."""
        Us = np.ascontiguousarray(Us, np.float32).reshape(-1, 44)
        n = Us.shape[0]
        if not 1 <= n <= self.batch:
            raise ValueError(f"a step takes 1 .. {self.batch} frames")
        slot = self.n_steps % 2
        self.n_steps += 1
        self.count[slot] = n             # a short batch (the tail of a frame loop) uses the front of the buffers
        packed = self.packed_ring[slot][:n]
        if self.streams:
            self.b.wait_for(self.free[slot])
        if self.windowed:
            rects = [self.b.frame_rect(Us[f]) for f in range(n)]
            self.window[slot] = (exchange_window([r for r, _ in rects], self.w, self.h, self.world, self.rpr), rects[0][1])
        if hasattr(self.b, "render_packed_batch"):
            self.b.render_packed_batch(Us, self.opts, packed)
        else:
            for f in range(n):
                self.b.render_packed(Us[f], self.opts, packed[f], False)
        if not self.collective:
            self.images = packed
            return
        if self.streams and getattr(self.b, "fif", 1) == 1:
            self.rendered[slot] = self.b.mark_batch(0)       # (with pipelines in flight the event is taken at exchange time)
        if self.prev is not None:
            self._exchange(self.prev, keep_latest=1)         # batch i-1, while batch i traverses
        self.prev = slot

    def _exchange(self, slot, keep_latest):
        n = self.count[slot]
        packed, gathered = self.packed_ring[slot][:n], self.gathered_ring[slot]
        if gathered is not None:
            gathered = gathered.view(-1)[:self.world * n * self.rpr * self.w].view(self.world * n, self.rpr, self.w)
        images = self.images[:n] if self.images is not None else None
        if self.streams:
            after = self.rendered[slot] if self.rendered[slot] is not None else self.b.mark_batch(keep_latest)
            self.rendered[slot] = None
            if self.windowed:
                window, fill = self.window[slot]
                self.free[slot] = self.b.exchange_batch_window(self.gathered_ring[slot], self.staging_ring[slot], packed, images, after,
                                                               self.root_only, self.w, self.h, self.world, window, fill)
            else:
                self.free[slot] = self.b.exchange_batch(gathered, packed, images, after, self.root_only,
                                                        self.w, self.h, self.world, self.rpr)
            return
        gather_stripes(gathered.view(-1, self.w) if gathered is not None else None, packed.reshape(-1, self.w), self.root_only)
        if self.has_frame:
            g = gathered.view(self.world, n, self.rpr, self.w)
            for f in range(n):
                self.b.deinterleave(g[:, f].contiguous().view(-1, self.w), images[f], self.w, self.h, self.world, self.rpr)

    def finish(self):
        if self.collective and self.prev is not None:
            self._exchange(self.prev, keep_latest=0)
            self.prev = None
        if hasattr(self.b, "finish"):
            self.b.finish()
