#!/usr/bin/env python3
"""Host cost of queueing one pipeline of 20 frames (rt_render_batch_device), and when the GPU finishes: where the bench's
wall time per pipeline minus its kernels' time goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

ctx = rt.Context(0)
ctx.set_mesh(rt.normalize_mesh(scenes.s1m())); ctx.build_lbvh()
W, H, B = 3840, 2160, 20
Us = np.stack([rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0)) for k in range(B)])
img = torch.zeros((B, H, W), dtype=torch.int32, device="cuda:0")
ptrs = [img[f].data_ptr() for f in range(B)]
opts = rt.render_opts(mode=1)
for _ in range(3):
    ctx.render_batch_device(Us, opts, ptrs)
ctx.synchronize()
for prof in (False, True):
    enq, tot, ker = [], [], []
    for _ in range(10):
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if prof:
            ctx.profile_begin()
        ctx.render_batch_device(Us, opts, ptrs)
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        if prof:
            ker.append(ctx.profile_end_phases()[0])
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"profile events {'on ' if prof else 'off'}: enqueue {np.median(enq):.3f} ms, until idle {np.median(tot):.3f} ms" +
          (f", kernels (events) {np.median(ker):.3f} ms" if ker else ""))
ctx.close()
