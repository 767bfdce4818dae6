#!/usr/bin/env python3
"""Per-rank frame / k_render time of each rank's stripe share of the cfg-3 frame on ONE GPU (hipEvents, one frame in flight),
with and without the frame-coherent scheduling hints: finds ranks whose share is slower than the others'."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import raytracer_amd as rt
from raytracer_amd import scenes

world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(0)
ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
scenes.bind_standin_surface(ctx, m)
W, H = 3840, 2160
U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
buf = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
for hints in (1, 0):
    ctx.set_coherent_scheduling(hints)
    for rank in range(world):
        o = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED, stripe_rows=16, stripe_rank=rank, stripe_world=world)
        for _ in range(10):
            ctx.render_device(U, o, d_rgba8=buf.data_ptr(), want_stats=False)
        ctx.synchronize()
        ctx.profile_begin()
        for _ in range(50):
            ctx.render_device(U, o, d_rgba8=buf.data_ptr(), want_stats=False)
        ctx.synchronize()
        fr, tr, n = ctx.profile_end()
        oc = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED | rt.FLAG_COUNT_WORK, stripe_rows=16, stripe_rank=rank, stripe_world=world)
        st = ctx.render_device(U, oc, d_rgba8=buf.data_ptr())
        print(f"hints={hints} N={world} rank {rank}: frame {fr:.4f} ms, k_render {tr:.4f} ms, rays {st['rays_primary'] + st['rays_shadow']}, "
              f"box tests {st['box_tests']}", flush=True)
ctx.close()
