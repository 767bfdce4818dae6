"""CPU-only, world_size 2 over gloo: the multi-GPU host logic of raytracer_amd/dist.py (BVH
replication + stripe sharding + packed all-gather + de-interleave) with the ORACLE standing in
for the GPU backend.  The gathered frame must equal the unsharded oracle frame on every rank."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Same interface as dist.GpuBackend, rendering this rank's stripes with the CPU oracle."""

    def __init__(self, oracle, bvh):
        self.o, self.bvh = oracle, bvh

    def alloc(self, rows, width):
        return torch.zeros((rows, width), dtype=torch.int32)

    def render_packed(self, U, opts, packed, want_stats=True):
        from raytracer_amd import dist as rdist
        h = int(U[37])
        ys = rdist.stripe_rows_of(h, opts.stripe_rank, opts.stripe_world) if opts.stripe_rows else np.arange(h)
        out, st = self.o.render(self.bvh["nodes"], self.bvh["tri_pos"], self.bvh["tri_nor"], U, mode=opts.mode,
                                gi_samples=opts.gi_samples, want=("rgba8",))
        img = out["rgba8"].view(np.int32)[:, :, 0]
        packed[:len(ys)] = torch.from_numpy(img[ys].copy())
        return st

    def deinterleave(self, gathered, image, width, height, world, rpr):
        from raytracer_amd import dist as rdist
        g = gathered.view(world, rpr, width)
        for r in range(world):
            ys = rdist.stripe_rows_of(height, r, world)
            image[torch.from_numpy(ys)] = g[r, :len(ys)]


class FakeCtx:
    """counts / export / adopt of the C ABI, on host memory."""

    def __init__(self, bvh=None):
        self.bvh, self.adopted, self.enc = bvh, None, 1

    def node_encoding(self):
        return self.enc

    def set_node_encoding(self, enc):
        self.enc = enc

    def counts(self):
        return self.bvh["tri_pos"].shape[0], self.bvh["nodes"].shape[0]

    def export_bvh_device(self, pn, pp, pq, N, T):
        for ptr, arr in ((pn, self.bvh["nodes"]), (pp, self.bvh["tri_pos"]), (pq, self.bvh["tri_nor"])):
            a = np.ascontiguousarray(arr, np.float32)
            ctypes.memmove(ptr, a.ctypes.data, a.nbytes)

    def tri_uv_device(self):
        uv = None if self.bvh is None else self.bvh.get("tri_uv")
        return (0, 0) if uv is None else (uv.ctypes.data, uv.nbytes)

    def export_tri_uv_device(self, dst, T):
        uv = self.bvh["tri_uv"]
        assert uv.nbytes == 32 * T
        ctypes.memmove(dst, uv.ctypes.data, uv.nbytes)

    def adopt_tri_uv_device(self, src, T):
        self.adopted_uv = np.frombuffer((ctypes.c_float * (8 * T)).from_address(src), np.float32).copy().reshape(T, 8)

    def adopt_bvh_device(self, pn, pp, pq, N, T):
        rd = lambda p, n: np.frombuffer((ctypes.c_float * n).from_address(p), np.float32).copy()
        self.adopted = dict(nodes=rd(pn, 8 * N).reshape(N, 8), tri_pos=rd(pp, 12 * T).reshape(T, 3, 4), tri_nor=rd(pq, 12 * T).reshape(T, 3, 4))


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracer_amd as rt
        from oracle import binding as oracle
        from raytracer_amd import dist as rdist
        mesh = rt.generate_test_scene()
        bvh = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx) if rank == 0 else None
        uv_ref = np.arange(16, dtype=np.float32).reshape(2, 8) / 16            # the surface records travel with the BVH
        if rank == 0:
            bvh["tri_uv"] = uv_ref.copy()
        ctx = FakeCtx(bvh)
        keep = rdist.replicate_bvh(ctx, rank, world, "cpu")           # BVH built on rank 0 only, broadcast to the rest
        assert rank == 0 or np.array_equal(ctx.adopted_uv, uv_ref)
        mine = bvh if rank == 0 else ctx.adopted
        assert mine is not None and len(keep) == (0 if rank == 0 else 1)
        w, h = 50, 70                                                   # not multiples of the 16-row stripes
        U = rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 3.0))
        U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
        frame = rdist.ShardedFrame(OracleBackend(oracle, mine), w, h, rank, world, rt.render_opts(mode=rt.MODE_SHADOW))
        assert frame.rpr == rt.packed_rows(h, 16, 0, world)
        frame.step(U)
        ref, _ = oracle.render(mine["nodes"], mine["tri_pos"], mine["tri_nor"], U, mode=1, want=("rgba8",))
        ok = np.array_equal(frame.image.numpy(), ref["rgba8"].view(np.int32)[:, :, 0]) and (ref["rgba8"][..., 0] > 0).any()
        # asynchronous, software-pipelined loop: frame i is de-interleaved by step i+1 (or finish()), two buffer pairs alternate
        cams = [(0.0, 0.0, 3.0), (0.3, 0.1, 2.5), (-0.2, 0.2, 2.0), (0.1, -0.3, 3.5)]
        refs = []
        for k, cam in enumerate(cams):
            Uk = rt.make_uniforms(w, h, cam_pos=cam)
            Uk[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
            refs.append(oracle.render(mine["nodes"], mine["tri_pos"], mine["tri_nor"], Uk, mode=1, want=("rgba8",))[0]["rgba8"].view(np.int32)[:, :, 0])
            frame.step(Uk, want_stats=False)
            if k >= 1:
                ok = ok and np.array_equal(frame.image.numpy(), refs[k - 1])
        frame.finish()
        ok = ok and np.array_equal(frame.image.numpy(), refs[-1]) and not np.array_equal(refs[-1], refs[-2])
        # exchange="root": the same loop, the frame arrives on rank 0 only (grouped send / recv instead of the all-gather)
        rframe = rdist.ShardedFrame(OracleBackend(oracle, mine), w, h, rank, world, rt.render_opts(mode=rt.MODE_SHADOW), exchange="root")
        ok = ok and ((rframe.image is None) == (rank != 0))
        for k, cam in enumerate(cams):
            Uk = rt.make_uniforms(w, h, cam_pos=cam)
            Uk[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
            rframe.step(Uk, want_stats=False)
            if k >= 1 and rank == 0:
                ok = ok and np.array_equal(rframe.image.numpy(), refs[k - 1])
        rframe.finish()
        if rank == 0:
            ok = ok and np.array_equal(rframe.image.numpy(), refs[-1])
        # batches: 2 frames per step through one exchange of [2, rows_per_rank, w] per rank (BatchedFrames); the frames of
        # batch i are complete after step i+1 or finish()
        for exchange in ("root", "all"):
            bf = rdist.BatchedFrames(OracleBackend(oracle, mine), w, h, rank, world, 2, rt.render_opts(mode=rt.MODE_SHADOW), exchange=exchange)
            has = rank == 0 or exchange == "all"
            ok = ok and ((bf.images is None) == (not has))
            Ub = []
            for cam in cams:
                Uk = rt.make_uniforms(w, h, cam_pos=cam)
                Uk[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
                Ub.append(Uk)
            bf.step(np.stack(Ub[0:2]))
            bf.step(np.stack(Ub[2:4]))
            if has:
                ok = ok and np.array_equal(bf.images[0].numpy(), refs[0]) and np.array_equal(bf.images[1].numpy(), refs[1])
            bf.finish()
            if has:
                ok = ok and np.array_equal(bf.images[0].numpy(), refs[2]) and np.array_equal(bf.images[1].numpy(), refs[3])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_sharded_frame():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    procs = [ctxmp.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_batch_sizes_are_even():
    from raytracer_amd import dist as rdist
    assert rdist.batch_sizes(20, 16) == [10, 10] and rdist.batch_sizes(20, 8) == [7, 7, 6] and rdist.batch_sizes(5, 16) == [5]
    assert rdist.batch_sizes(16, 16) == [16] and rdist.batch_sizes(17, 16) == [9, 8] and rdist.batch_sizes(0, 16) == []
    for n in range(1, 70):
        for b in (1, 3, 8, 16):
            sz = rdist.batch_sizes(n, b)
            assert sum(sz) == n and max(sz) <= b and max(sz) - min(sz) <= 1 and len(sz) == -(-n // b)


def test_stripe_ownership_partition():
    from raytracer_amd import dist as rdist
    for h in (1, 16, 33, 150, 2160):
        for world in (1, 2, 4, 8):
            ys = np.concatenate([rdist.stripe_rows_of(h, r, world) for r in range(world)])
            assert np.array_equal(np.sort(ys), np.arange(h))
            assert rdist.rows_per_rank(h, world) >= max(len(rdist.stripe_rows_of(h, r, world)) for r in range(world))


def test_exchange_window_covers_every_ranks_share_of_the_rectangles():
    """dist.exchange_window (pure host logic): the window must hold, on EVERY rank, all packed rows / columns of the union of the
    frames' rectangles, start on a 16-byte column, and stay inside the packed buffers."""
    from raytracer_amd import dist as rdist
    rng = np.random.default_rng(11)
    for _ in range(400):
        w, h, world = int(rng.integers(4, 400)), int(rng.integers(1, 300)), int(rng.integers(1, 9))
        rpr = -(-(-(-h // 16)) // world) * 16          # an upper bound of rt_packed_rows(h, 16, 0, world), as the callers allocate at least
        rects = []
        for _f in range(int(rng.integers(1, 5))):
            x0, y0 = int(rng.integers(-5, w + 5)), int(rng.integers(-5, h + 5))
            rects.append((x0, y0, x0 + int(rng.integers(-3, w)), y0 + int(rng.integers(-3, h))))
        col0, n_cols, g0, n_rows = rdist.exchange_window(rects, w, h, world, rpr)
        live = [r for r in rects if min(r[2], w - 1) >= max(r[0], 0) and min(r[3], h - 1) >= max(r[1], 0)]      # on screen
        if not live:
            assert (n_cols, n_rows) == (0, 0)
            continue
        assert col0 % 4 == 0 and col0 + n_cols <= w and g0 * 16 + n_rows <= rpr and n_rows > 0 and n_cols > 0
        for (x0, y0, x1, y1) in live:
            for y in range(max(y0, 0), min(y1, h - 1) + 1):
                stripe = y // 16
                p = (stripe // world) * 16 + y % 16      # the packed row of image row y on its owner, rank stripe % world
                assert g0 * 16 <= p < g0 * 16 + n_rows, (rects, w, h, world)
            if max(x0, 0) <= min(x1, w - 1):
                assert col0 <= max(x0, 0) and min(x1, w - 1) < col0 + n_cols
