#!/usr/bin/env python3
"""What a batch buys: ms per frame (wall clock, asynchronous loop) and k_render time per frame (hipEvents) of
  per-frame rendering (1 and 3 frames in flight)  vs  rt_render_batch_device with 2 / 4 / 8 frames per pipeline
for S70k 1080p (cfg 1-2), cfg 3 at 4K and a 1/8 stripe share of it.  usage: batch_gain.py [--moving]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
from raytracer_amd import scenes

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
moving = "--moving" in sys.argv


def case(name, mesh, W, H, stripe=None, stripe_rows=16):
    ctx = rt.Context(0)
    ctx.set_mesh(mesh); ctx.set_surface(mesh); ctx.build_lbvh()
    ctx.set_coherent_scheduling("--nohints" not in sys.argv)
    scenes.bind_standin_surface(ctx, mesh)
    so = dict(stripe_rows=stripe_rows, stripe_rank=stripe[0], stripe_world=stripe[1]) if stripe else {}
    o = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED, **so)
    oc = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED | rt.FLAG_COUNT_WORK, **so)
    rows = rt.packed_rows(H, stripe_rows, *stripe) if stripe else H
    views = [rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, (1.0 * k if moving else 0.0), 0.0)) for k in range(8)]
    bufs = torch.empty((32, rows, W), dtype=torch.int32, device=dev)
    c = ctx.render_device(views[0], oc, d_rgba8=bufs[0].data_ptr())
    byts = 32 * c["box_tests"] + 48 * c["tri_tests"] + 48 * c["hit_updates"] + 4 * c["pixels"] - 36 * (c["pixels"] - c["pixels_traced"])
    n = 240
    print(f"{name:28s} traced pixels {c['pixels_traced']}, rays {c['rays_primary'] + c['rays_shadow']}, algorithmic bytes {byts / 1e6:.1f} MB", flush=True)

    def report(tag, e, tr_ms, launches, frames=n):
        per_frame = tr_ms * launches / frames
        print(f"{name:28s} {tag:22s} {e / frames * 1e3:8.4f} ms/frame  k_render {per_frame:7.4f} ms/frame  frac {byts / (per_frame * 1e-3) / 1e9 / 8000:6.3f}", flush=True)

    for fif in (1, 3):
        ctx.set_frames_in_flight(fif)
        for i in range(16):
            ctx.render_device(views[i % 8], o, d_rgba8=bufs[i % 8].data_ptr(), want_stats=False)
        ctx.frames_join(0); ctx.synchronize()
        ctx.profile_begin()
        t0 = time.perf_counter()
        for i in range(n):
            ctx.render_device(views[i % 8], o, d_rgba8=bufs[i % 8].data_ptr(), want_stats=False)
        ctx.frames_join(0); ctx.synchronize()
        e = time.perf_counter() - t0
        _, tr, k = ctx.profile_end()
        report(f"per frame, {fif} in flight", e, tr, k)
    for fif, sizes in ((1, (2, 4, 8, 16, 20, 32)), (2, (8, 16))):
      ctx.set_frames_in_flight(fif)
      for B in sizes:
          ptrs = [bufs[f].data_ptr() for f in range(B)]
          Ub = [np.stack([views[(i0 + f) % 8] for f in range(B)]) for i0 in range(0, max(8, B), B)]
          for i in range(4):
              ctx.render_batch_device(Ub[i % len(Ub)], o, ptrs)
          (ctx.frames_join(0), ctx.synchronize())
          ctx.profile_begin()
          t0 = time.perf_counter()
          for i in range(n // B):
              ctx.render_batch_device(Ub[i % len(Ub)], o, ptrs)
          (ctx.frames_join(0), ctx.synchronize())
          e = time.perf_counter() - t0
          _, tr, k = ctx.profile_end()
          report(f"batch of {B}" + (f", {fif} in flight" if fif > 1 else ""), e, tr, k, (n // B) * B)

    ctx.set_frames_in_flight(1)
    ctx.close()


s70k = rt.normalize_mesh(scenes.s70k())
s1m = rt.normalize_mesh(scenes.s1m())
if "--bands" in sys.argv:      # 16-row round-robin stripes vs contiguous bands of rows (272 = 2160 / 8 rounded up to a tile multiple)
    for r in (0, 2, 3, 4):
        case(f"S1M 4K, stripes, rank {r}/8", s1m, 3840, 2160, stripe=(r, 8))
        case(f"S1M 4K, band, rank {r}/8", s1m, 3840, 2160, stripe=(r, 8), stripe_rows=272)
    sys.exit(0)
case("S70k 1920x1080", s70k, 1920, 1080)
case("S1M 3840x2160", s1m, 3840, 2160)
case("S1M 4K, rank 3 of 8", s1m, 3840, 2160, stripe=(3, 8))
case("S1M 4K, rank 0 of 2", s1m, 3840, 2160, stripe=(0, 2))
