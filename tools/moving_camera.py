#!/usr/bin/env python3
"""Does frame-coherent scheduling survive camera motion?  cfg 3 scene, the camera orbits the mesh by a fixed angle per frame
(sliders rotY, as a user dragging the reference's rotation slider); traversal-kernel time with and without hints."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
from raytracer_amd import scenes

ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m)
ctx.set_surface(m)
ctx.build_lbvh()
scenes.bind_standin_surface(ctx, m)
W, H = 3840, 2160
img = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
opts = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
for deg in (0.0, 0.25, 1.0, 4.0, 15.0):
    row = []
    for on in (False, True):
        ctx.set_coherent_scheduling(on)
        ts = []
        for k in range(24):
            U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, deg * k, 0.0))
            st = ctx.render_device(U, opts, d_rgba8=img.data_ptr())
            if k >= 4:
                ts.append(st["ms_traverse"])
        row.append(float(np.mean(ts)))
    print(f"rotY step {deg:5.2f} deg/frame: k_render {row[0]:.3f} ms without hints, {row[1]:.3f} ms with ({(1 - row[1] / row[0]) * 100:+.1f} %)", flush=True)
ctx.close()
