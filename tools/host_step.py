import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes, dist as rdist
W,H,K=3840,2160,20
m=rt.normalize_mesh(scenes.s1m())
ctx=rt.Context(0); ctx.set_mesh(m); ctx.set_surface(m); ctx.build_lbvh()
ctx.set_materials(scenes.STANDIN_MATERIALS); ctx.set_texture(0, scenes.procedural_texture(2048,0)); ctx.set_texture(1, scenes.procedural_texture(1024,1))
dev=torch.device('cuda:0')
opts=rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
views=[rt.make_uniforms(W,H,cam_pos=(0,0,1.3),rot_deg=(0.0,1.0*k,0.0)) for k in range(K)]
bf=rdist.BatchedFrames(rdist.GpuBackend(ctx,dev,frames_in_flight=1),W,H,0,1,32,opts,exchange='root')
for rep in range(4):
    bf.finish(); torch.cuda.synchronize()
    t0=time.perf_counter()
    Us=np.stack(views)
    t1=time.perf_counter()
    ctx.profile_begin()
    bf.step(Us)
    t2=time.perf_counter()
    bf.finish(); torch.cuda.synchronize()
    t3=time.perf_counter()
    pm,tm,sm,n=ctx.profile_end_phases()
    print(f"stack {1e3*(t1-t0):.3f} ms, step (host enqueue) {1e3*(t2-t1):.3f} ms, wait {1e3*(t3-t2):.3f} ms, total {1e3*(t3-t0):.3f}; pipeline events {pm:.3f} ms")
# raw call
out=torch.zeros((K,H,W),dtype=torch.int32,device=dev)
ptrs=[out[k].data_ptr() for k in range(K)]
for rep in range(3):
    ctx.synchronize(); torch.cuda.synchronize()
    t0=time.perf_counter(); ctx.render_batch_device(np.stack(views), opts, ptrs); t1=time.perf_counter(); ctx.synchronize(); t2=time.perf_counter()
    print(f"raw render_batch_device enqueue {1e3*(t1-t0):.3f} ms, wait {1e3*(t2-t1):.3f}, total {1e3*(t2-t0):.3f}")
