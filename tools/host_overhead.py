#!/usr/bin/env python3
"""Host-side cost of one frame-loop step (kernel launches, events, collective, de-interleave) measured with a tiny
frame so that the GPU is never the bottleneck: an upper bound on frames/s per rank in the multi-GPU loop.
  python3 tools/host_overhead.py            (one GPU; world-1 RCCL group, gather leg forced on)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import raytracer_amd as rt
from raytracer_amd import dist as rdist, scenes


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29555", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    ctx = rt.Context(0)
    m = rt.normalize_mesh(scenes.s70k())
    ctx.set_mesh(m)
    ctx.build_lbvh()
    dev = torch.device("cuda:0")
    for (w, h) in ((3840, 272), (3840, 544), (3840, 2160)):
        U = rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 1.3))
        for coll, fif in ((False, 1), (False, 2), (False, 3), (False, 4), (True, 2), (True, 3), (True, 4)):
            f = rdist.ShardedFrame(rdist.GpuBackend(ctx, dev, frames_in_flight=fif), w, h, 0, 1, rt.render_opts(mode=rt.MODE_SHADOW), force_collective=coll)
            for _ in range(20):
                f.step(U, want_stats=False)
            f.finish()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                f.step(U, want_stats=False)
            t_enq = time.perf_counter() - t0
            f.finish()
            t_all = time.perf_counter() - t0
            print(f"{w}x{h} collective={coll} frames_in_flight={fif}: enqueue {t_enq / n * 1e6:.1f} us/step, drained {t_all / n * 1e6:.1f} us/step", flush=True)
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
