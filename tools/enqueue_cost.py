#!/usr/bin/env python3
"""Host time to ENQUEUE one frame of a 1/8 stripe share (render + exchange + de-interleave) vs the GPU's time per frame:
is the N = 8 frame loop bound by the host?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import raytracer_amd as rt
from raytracer_amd import dist as rdist, scenes

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29558", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
scenes.bind_standin_surface(ctx, m)
dev = torch.device("cuda:0")
W, H, world = 3840, 2160, 8
U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
base = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
for exchange, coll in (("root", True), ("all", True), ("none", False)):
    f = rdist.ShardedFrame(rdist.GpuBackend(ctx, dev, frames_in_flight=3), W, H, 0, 1, base, force_collective=coll, exchange=exchange if coll else "all")
    rpr = rdist.rows_per_rank(H, world)
    f.rpr = rpr
    f.opts = rt.render_opts(mode=base.mode, flags=base.flags, stripe_rows=16, stripe_rank=3, stripe_world=world)
    f.packed_ring = [f.b.alloc(rpr, W) for _ in f.packed_ring]
    if coll:
        f.gathered_ring = [f.b.alloc(rpr, W) for _ in f.gathered_ring]
        f.image = f.b.alloc(rpr, W); f.h = rpr
    else:
        f.packed = f.packed_ring[0]
    for _ in range(30):
        f.step(U, want_stats=False)
    f.finish(); torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        f.step(U, want_stats=False)
    t1 = time.perf_counter()
    f.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    # pure host cost: the same calls with the GPU idle ahead of us is not separable; report loop time and total
    print(f"exchange {exchange}: enqueue loop {(t1 - t0) / n * 1e6:.1f} us/frame, total {(t2 - t0) / n * 1e6:.1f} us/frame", flush=True)
# the render call alone (no exchange), one frame in flight so the host never blocks on a full queue
ctx.set_frames_in_flight(1)
o = rt.render_opts(mode=base.mode, flags=base.flags, stripe_rows=16, stripe_rank=3, stripe_world=world)
buf = torch.zeros((rdist.rows_per_rank(H, world), W), dtype=torch.int32, device=dev)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ctx.render_device(U, o, d_rgba8=buf.data_ptr(), want_stats=False)
t1 = time.perf_counter()
ctx.synchronize()
print(f"rt_render_device alone: {(t1 - t0) / 50 * 1e6:.1f} us per call to enqueue (50 calls, queue not full)")
dist.destroy_process_group(); ctx.close()
