#!/usr/bin/env python3
"""Work counters of one cfg-3 frame (the kernel's own, equal to the oracle's): box / triangle tests, hit updates, rays, and
the number of 16-byte lane requests the WIDE64 walk issues for them (4 per internal step, 3 per triangle step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "s1m"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
m = rt.normalize_mesh(scenes.s1m() if scene == "s1m" else scenes.s70k())
ctx = rt.Context(0)
ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
ctx.set_materials(scenes.STANDIN_MATERIALS); ctx.set_texture(0, scenes.procedural_texture(2048, 0)); ctx.set_texture(1, scenes.procedural_texture(1024, 1))
import torch
out = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
for mode in (rt.MODE_PRIMARY, rt.MODE_SHADOW):
    U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
    c = ctx.render_device(U, rt.render_opts(mode=mode, flags=rt.FLAG_COUNT_WORK | rt.FLAG_TEXTURED), d_rgba8=out.data_ptr())
    print(mode, {k: int(v) if isinstance(v, int) else v for k, v in c.items()})
ctx.close()
