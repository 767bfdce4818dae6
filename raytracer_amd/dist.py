"""Multi-GPU host logic: one process per GPU, torch.distributed (RCCL) for
the two exchange steps the path really has —
  * BVH replication: built once on rank 0, the three flat buffers are broadcast over xGMI;
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

    def finish(self):
        self.ctx.synchronize()
        torch.cuda.current_stream(self.device).synchronize()


def gather_stripes(gathered, packed, root_only):
    """The frame's one exchange step.  gathered: [world * rows_per_rank, width] (rank-major), or None off the root."""
    if not root_only:
        dist.all_gather_into_tensor(gathered, packed)
    elif dist.get_rank() == 0:
        dist.gather(packed, list(gathered.view(dist.get_world_size(), *packed.shape).unbind(0)), dst=0)
    else:
        dist.gather(packed, None, dst=0)


def replicate_bvh(ctx, rank, world, device):
    """Broadcast rank 0's flat BVH (nodes, triPos, triNor, and the BVH-order surface records when the scene has them)
    and adopt it on every other rank.  Returns the tensors that back the adopted BVH (keep them alive).
    Materials and textures are small host-side descriptions: every rank sets them itself."""
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == 0:
        T, N = ctx.counts()
        uv_ptr, uv_bytes = ctx.tri_uv_device() if hasattr(ctx, "tri_uv_device") else (0, 0)
        meta[0], meta[1], meta[2], meta[3] = T, N, ctx.node_encoding(), 1 if uv_bytes else 0
    dist.broadcast(meta, 0)
    T, N, enc, has_uv = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
    bufs = [torch.empty(32 * N, dtype=torch.uint8, device=device), torch.empty(48 * T, dtype=torch.uint8, device=device),
            torch.empty(48 * T, dtype=torch.uint8, device=device)]
    if rank == 0:
        ctx.export_bvh_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), N, T)
    for b in bufs:                      # three large broadcasts (32 B x N, 48 B x T, 48 B x T)
        dist.broadcast(b, 0)
    uv = None
    if has_uv:                          # 32 B x T surface records
        uv = torch.empty(32 * T, dtype=torch.uint8, device=device)
        if rank == 0:
            ctx.export_tri_uv_device(uv.data_ptr(), T)
        dist.broadcast(uv, 0)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if rank != 0:
        ctx.set_node_encoding(enc)          # v1 / v2 node records travel as the same 32-byte slots
        ctx.adopt_bvh_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), N, T)
        if uv is not None:
            ctx.adopt_tri_uv_device(uv.data_ptr(), T)   # copied: `uv` need not be kept
    return bufs


class ShardedFrame:
    """One frame sharded as image-space stripes over `world` ranks and gathered on every rank.

    Asynchronous frame loops (step(want_stats=False)) are software-pipelined: step i queues the render of frame i
    first and only then the all-gather + de-interleave of frame i-1, so with two frames in flight (GpuBackend) the
    xGMI exchange and the tail of frame i-1 overlap the traversal of frame i.  Two packed / gathered buffer pairs
    alternate; frame i+2 reuses pair i only after all-gather(i) (a frame starts behind everything queued on the
    context stream before it)."""

    DEPTH = 2      # ring of packed / gathered buffers and frames kept un-gathered (more with GpuBackend(frames_in_flight > 2))

    def __init__(self, backend, width, height, rank, world, base_opts=None, force_collective=False, exchange="all"):
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

    def _exchange(self, slot, keep_latest):
        packed, gathered = self.packed_ring[slot], self.gathered_ring[slot]
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
