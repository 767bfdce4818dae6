#!/usr/bin/env python3
"""Analysis aid (needs a library built with `make lib EXTRA=-DRT_TIMELINE`): when do the waves of one k_render launch
start their last chunk and finish?  Prints the distribution relative to the launch's first start."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

mode = dict(primary=0, shadow=1, pt=2)[sys.argv[1] if len(sys.argv) > 1 else "shadow"]
ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m); ctx.build_lbvh()
W, H = 3840, 2160
U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
import torch
img = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
opts = rt.render_opts(mode=mode, gi_samples=16)
moving = len(sys.argv) > 2 and sys.argv[2] == "moving"      # the bench's per-frame leg: the model turns 1 degree per frame
for k in range(12 if moving else 3):
    if moving:
        U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0))
    st = ctx.render_device(U, opts, d_rgba8=img.data_ptr())
L = C.CDLL(_abi.LIB_PATH)
buf = np.zeros((4096 * 4, 4), np.uint64)
rc = L.rt_debug_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0
b = buf[buf[:, 2] > 0].astype(np.int64)
t0 = b[:, 0].min()
tick = 1e-2   # wall_clock64 runs at 100 MHz: 10 ns per tick -> us
start, last, end, n = (b[:, 0] - t0) * tick, (b[:, 1] - t0) * tick, (b[:, 2] - t0) * tick, b[:, 3] & 0xFFFFFFFF
longest = (b[:, 3] >> 32) * tick            # the wave's longest chunk
print(f"waves {len(b)}  kernel_ms(events) {st['ms_traverse']:.3f}  span {end.max():.1f} us")
for name, v in (("start", start), ("last chunk start", last), ("end", end), ("last chunk duration", end - last), ("longest chunk", longest), ("chunks/wave", n.astype(float))):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100])
    print(f"{name:22s} min {q[0]:8.1f}  p10 {q[1]:8.1f}  p50 {q[2]:8.1f}  p90 {q[3]:8.1f}  p99 {q[4]:8.1f}  max {q[5]:8.1f}")
# how many waves are still running at time t
for frac in (0.5, 0.7, 0.8, 0.85, 0.9, 0.95, 0.98):
    t = frac * end.max()
    print(f"  t = {t:7.1f} us ({int(frac * 100)} %): {(end > t).sum():5d} waves still running")
ctx.close()
