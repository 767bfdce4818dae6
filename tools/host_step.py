#!/usr/bin/env python3
"""Host-side cost of enqueueing one batched pipeline the way bench.py does (cProfile of BatchedFrames.step)."""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes, dist as rdist
W, H, K = 3840, 2160, 20
m = rt.normalize_mesh(scenes.s1m())
ctx = rt.Context(0); ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
ctx.set_materials(scenes.STANDIN_MATERIALS); ctx.set_texture(0, scenes.procedural_texture(2048, 0)); ctx.set_texture(1, scenes.procedural_texture(1024, 1))
dev = torch.device('cuda:0')
opts = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
views = [rt.make_uniforms(W, H, cam_pos=(0, 0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0)) for k in range(K)]
frame = rdist.ShardedFrame(rdist.GpuBackend(ctx, dev, frames_in_flight=1), W, H, 0, 1, opts, exchange='root')
bf = rdist.BatchedFrames(frame.b, W, H, 0, 1, 32, opts, exchange='root')
def one():
    ctx.profile_begin()
    bf.step(np.stack([views[i % K] for i in range(K)]))
for rep in range(3):
    one(); bf.finish(); torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); one(); pr.disable(); t1 = time.perf_counter()
bf.finish(); torch.cuda.synchronize()
print(f"enqueue {1e3 * (t1 - t0):.3f} ms (under cProfile)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
