#!/usr/bin/env python3
"""Host-side time of each leg of ShardedFrame.step (world-1 RCCL group, gather leg forced on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import raytracer_amd as rt
from raytracer_amd import dist as rdist, scenes

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29556", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s70k())
ctx.set_mesh(m); ctx.build_lbvh()
dev = torch.device("cuda:0")
w, h = 3840, 272
U = rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 1.3))
b = rdist.GpuBackend(ctx, dev, frames_in_flight=2)
f = rdist.ShardedFrame(b, w, h, 0, 1, rt.render_opts(mode=rt.MODE_SHADOW), force_collective=True)
for _ in range(20):
    f.step(U, want_stats=False)
f.finish()
acc = dict(render=0.0, wait=0.0, gather=0.0, event=0.0, deint=0.0)
n = 300
pk, ga = f.packed_ring[0], f.gathered_ring[0]
for _ in range(n):
    t0 = time.perf_counter(); b.render_packed(U, f.opts, pk, False)
    t1 = time.perf_counter(); b.before_gather()
    t2 = time.perf_counter(); dist.all_gather_into_tensor(ga, pk)
    t3 = time.perf_counter(); ev = b.after_gather()
    t4 = time.perf_counter(); b.deinterleave(ga, f.image, w, h, 1, f.rpr, after=ev)
    t5 = time.perf_counter()
    for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += d
f.finish()
print({k: round(v / n * 1e6, 1) for k, v in acc.items()}, "us per step")
# the raw collective alone, and async_op
t0 = time.perf_counter()
for _ in range(n):
    dist.all_gather_into_tensor(ga, pk)
torch.cuda.synchronize()
print("all_gather alone", round((time.perf_counter() - t0) / n * 1e6, 1), "us")
t0 = time.perf_counter()
for _ in range(n):
    dist.all_gather_into_tensor(ga, pk, async_op=True)
torch.cuda.synchronize()
print("all_gather async_op", round((time.perf_counter() - t0) / n * 1e6, 1), "us")
small = torch.zeros(1024, dtype=torch.int32, device=dev); big = torch.zeros(1024, dtype=torch.int32, device=dev)
t0 = time.perf_counter()
for _ in range(n):
    dist.all_gather_into_tensor(big, small)
torch.cuda.synchronize()
print("all_gather 4 KB", round((time.perf_counter() - t0) / n * 1e6, 1), "us")
dist.destroy_process_group(); ctx.close()
