#!/usr/bin/env python3
"""Projection of the N-GPU frame loop from ONE GPU: rank r's share of the cfg-3 frame (16-row stripes, stripe s -> rank s % N)
rendered in the real loop (frames in flight, gather leg forced on through a world-1 RCCL group, de-interleave), timed per
frame.  The N-GPU rate is bounded by the slowest rank: projected speed-up = t(N=1) / max_r t_r(N).  What this cannot show:
the xGMI transfer itself (33 MB / N per rank and frame, << 0.1 ms) and cross-rank skew."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import raytracer_amd as rt
from raytracer_amd import dist as rdist, scenes

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29557", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
scenes.bind_standin_surface(ctx, m)
dev = torch.device("cuda:0")
W, H = 3840, 2160
U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
base = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)


def run_batched(world, rank, batch, n=200):
    """The N > 1 loop of bench.py: `batch` frames per pipeline, one exchange per batch on a side stream."""
    f = rdist.BatchedFrames(rdist.GpuBackend(ctx, dev, frames_in_flight=int(os.environ.get("PROJ_BATCH_FIF", "1"))), W, H, 0, 1, batch, base,
                            force_collective=True, window_exchange=False)   # bench.py at N > 1: one batch in flight (PROJ_BATCH_FIF=2: two);
                                                         # whole stripes are exchanged here (the faked 1-rank image has no frame rectangle)
    rpr = rdist.rows_per_rank(H, world)
    f.rpr = rpr
    f.opts = rt.render_opts(mode=base.mode, flags=base.flags, stripe_rows=16, stripe_rank=rank, stripe_world=world)
    alloc3 = lambda k, rows: f.b.alloc(k * rows, W).view(k, rows, W)
    f.packed_ring = [alloc3(batch, rpr) for _ in range(2)]
    f.gathered_ring = [alloc3(batch, rpr) for _ in range(2)]
    f.images = alloc3(batch, rpr)
    f.h = rpr                        # de-interleave of a 1-rank "image" of rpr rows
    Ub = np.stack([U] * batch)
    for _ in range(8):
        f.step(Ub)
    f.finish()
    torch.cuda.synchronize()
    if os.environ.get("PROJ_PATTERN") == "bench":      # bench.py's timed region at N > 1: K = 20 steps as 10 + 10, then the fence; median of 30
        U10 = np.stack([U] * 10)
        times = []
        for _ in range(30):
            f.step(U10); f.finish(); torch.cuda.synchronize()          # (the warm-up batch before the timed region)
            t0 = time.perf_counter()
            f.step(U10); f.step(U10)
            f.finish()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 20 * 1e3)
        ctx.set_frames_in_flight(1)
        return sorted(times)[len(times) // 2]
    chunks = []
    nb = max(1, n // 4 // batch)
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(nb):
            f.step(Ub)
        f.finish()
        torch.cuda.synchronize()
        chunks.append((time.perf_counter() - t0) / (nb * batch) * 1e3)
    ctx.set_frames_in_flight(1)
    return sorted(chunks)[1]


def run(world, rank, fif, n=200):
    f = rdist.ShardedFrame(rdist.GpuBackend(ctx, dev, frames_in_flight=fif), W, H, 0, 1, base, force_collective=world > 1, window_exchange=False)
    if world > 1:                    # this rank's stripes, packed; the world-1 group gathers just them
        rpr = rdist.rows_per_rank(H, world)
        f.rpr = rpr
        f.opts = rt.render_opts(mode=base.mode, flags=base.flags, stripe_rows=16, stripe_rank=rank, stripe_world=world)
        f.packed_ring = [f.b.alloc(rpr, W) for _ in f.packed_ring]
        f.gathered_ring = [f.b.alloc(rpr, W) for _ in f.gathered_ring]
        f.image = f.b.alloc(rpr, W)
        f.h = rpr                    # de-interleave of a 1-rank "image" of rpr rows
    for _ in range(20):
        f.step(U, want_stats=False)
    f.finish()
    torch.cuda.synchronize()
    # timed in 4 chunks, median chunk reported: a one-off host stall (~35 ms, seen once per process about a second after
    # start, whatever was running: hipEvent times of the frames' kernels are unaffected) would otherwise count as 0.17 ms/frame
    ctx.profile_begin()
    chunks = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(n // 4):
            f.step(U, want_stats=False)
        f.finish()
        torch.cuda.synchronize()
        chunks.append((time.perf_counter() - t0) / (n // 4) * 1e3)
    dt = sorted(chunks)[1]
    fr, tr, _ = ctx.profile_end()
    if "--verbose" in sys.argv:
        print(f"   world {world} rank {rank}: {dt:.4f} ms/frame; frame kernels {fr:.4f} ms, k_render {tr:.4f} ms (hipEvents, overlapped frames inflate both); "
              f"packed @ {[hex(t.data_ptr()) for t in f.packed_ring]}", flush=True)
    return dt


order = (lambda w: range(w)) if "--reverse" not in sys.argv else (lambda w: reversed(range(w)))
BATCH = int(os.environ.get("PROJ_BATCH", "16"))       # frames per pipeline, as bench.py --batch; 1: one frame per pipeline with PROJ_FIF frames in flight (the round-1 loop)
t1 = run_batched(1, 0, BATCH) if BATCH > 1 else run(1, 0, 1)
print(f"N=1 ({BATCH} frames per pipeline, as bench.py): {t1:.4f} ms/frame" if BATCH > 1 else f"N=1 (one frame per pipeline, one in flight): {t1:.4f} ms/frame", flush=True)
worlds = (2, 4, 8) if "--worlds" not in sys.argv else tuple(int(v) for v in sys.argv[sys.argv.index("--worlds") + 1].split(","))
for world in worlds:
    ts = {r: (run_batched(world, r, BATCH) if BATCH > 1 else run(world, r, int(os.environ.get("PROJ_FIF", "3")))) for r in order(world)}
    ts = [ts[r] for r in range(world)]
    print(f"N={world}: per-rank ms/frame {[round(t, 4) for t in ts]} -> slowest {max(ts):.4f}, projected speed-up {t1 / max(ts):.2f}x "
          f"({t1 / max(ts) / world * 100:.0f} % of linear)", flush=True)
dist.destroy_process_group()
ctx.close()
