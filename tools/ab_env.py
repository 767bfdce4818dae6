#!/usr/bin/env python3
"""A/B of environment-tunable kernel parameters on the bench's cfg-3 workload (moving camera, 20 views): k_render ms per frame
in 20-frame pipelines and with one frame per pipeline, for each setting of the variables given.
usage: ab_env.py VAR=a,b,c [VAR2=x,y] ...   (cartesian product; '-' = unset)"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

W, H, K = 3840, 2160, 20
mode = {"primary": rt.MODE_PRIMARY, "shadow": rt.MODE_SHADOW, "pt": rt.MODE_PT}[os.environ.get("AB_MODE", "shadow")]
m = rt.normalize_mesh(scenes.s1m() if os.environ.get("AB_SCENE", "s1m") == "s1m" else scenes.s70k())
ctx = rt.Context(0)
ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
ctx.set_materials(scenes.STANDIN_MATERIALS); ctx.set_texture(0, scenes.procedural_texture(2048, 0)); ctx.set_texture(1, scenes.procedural_texture(1024, 1))
views = [rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0)) for k in range(K)]
out = torch.zeros((K, H, W), dtype=torch.int32, device="cuda:0")
opts = rt.render_opts(mode=mode, gi_samples=int(os.environ.get("AB_SPP", "32")), flags=rt.FLAG_TEXTURED)
ref = None
vars_ = [a.split("=", 1) for a in sys.argv[1:]]
for combo in itertools.product(*[v.split(",") for _, v in vars_]):
    for (name, _), val in zip(vars_, combo):
        if val == "-": os.environ.pop(name, None)
        else: os.environ[name] = val
    res = []
    for per_frame in (False, True):
        best = 1e9
        for rep in range(4):
            ctx.profile_begin()
            if per_frame:
                for k in range(K): ctx.render_device(views[k], opts, d_rgba8=out[k].data_ptr(), want_stats=False)
            else:
                ctx.render_batch_device(np.stack(views), opts, [out[k].data_ptr() for k in range(K)])
            _, tr, n = ctx.profile_end()
            if rep: best = min(best, tr * n / K)
        res.append(best)
    img = out.cpu().numpy()
    same = "ref" if ref is None else ("identical" if np.array_equal(img, ref) else "DIFFERENT")
    if ref is None: ref = img
    print(" ".join(f"{n}={v}" for (n, _), v in zip(vars_, combo)), f": batch {res[0]:.4f} ms/frame, per-frame {res[1]:.4f} ms/frame, images {same}", flush=True)
ctx.close()
