"""GPU: contexts come and go without leaving anything behind.  Forty create -> upload -> build -> render (per frame, batched,
frames in flight, PT split, no-alloc entry point, textures, cache file round trip) -> rt_destroy cycles in one process: free
device memory and the process's resident set must end where they were after the first cycles (a leaked scratch buffer, pixel
list or texture would show as tens of MB per cycle, a leaked stream or event as host growth)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def _cycle(rt, scenes, torch, mesh, tmp_path, k):
    ctx = rt.Context(0)
    try:
        ctx.set_node_encoding(rt.NODES_V2 if k % 2 else rt.NODES_V1)
        ctx.set_mesh(mesh)
        ctx.set_surface(mesh)
        ctx.build_lbvh()
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        ctx.set_texture(0, scenes.procedural_texture(512, 0))
        ctx.set_texture(1, scenes.procedural_texture(256, 1))
        w, h = 640 + 16 * (k % 3), 360
        U = rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 0.9))
        tex = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
        out, st = ctx.render(U, tex)
        assert st["pixels"] == w * h and (out["hit_tri"] != 0xFFFFFFFF).mean() > 0.03
        bufs = [torch.zeros((4, h, w), dtype=torch.int32, device="cuda:0") for _ in range(3)]
        ptrs = lambda b: [b[f].data_ptr() for f in range(4)]
        ctx.render_batch_device(np.stack([U] * 4), tex, ptrs(bufs[0]), ptrs(bufs[1]), ptrs(bufs[2]))
        ctx.set_frames_in_flight(3)
        for f in range(6):
            ctx.render_device(U, tex, bufs[0][f % 4].data_ptr(), want_stats=False)
        ctx.render_batch_device(np.stack([U] * 3), tex, ptrs(bufs[0])[:3])
        ctx.set_frames_in_flight(1)
        ctx.render_device(U, rt.render_opts(mode=rt.MODE_PT, gi_samples=4, gi_bounces=2), bufs[0][0].data_ptr(), want_stats=False)
        ctx.render_device_noalloc(U, tex, bufs[0][1].data_ptr())
        ctx.synchronize()
        assert np.array_equal(bufs[0][1].cpu().numpy().view(np.uint8).reshape(h, w, 4), out["rgba8"])
        path = str(tmp_path / "cycle.bvh")
        uv = ctx.tri_uv()
        ctx.save_bvh(path)
        ctx.load_bvh(path)                               # (the cache file is the reference's: tree + triangles, no surface records)
        ctx.set_tri_uv(uv)
        ctx.render_device(U, tex, bufs[0][2].data_ptr(), want_stats=False)
        ctx.synchronize()
        assert np.array_equal(bufs[0][2].cpu().numpy().view(np.uint8).reshape(h, w, 4), out["rgba8"])
    finally:
        ctx.close()


def test_forty_context_lifetimes_leave_nothing_behind(rt, tmp_path):
    import torch
    from raytracer_amd import scenes
    mesh = rt.normalize_mesh(scenes.s70k())
    for k in range(3):                                   # allocator pools, code objects, torch's cache: grown once
        _cycle(rt, scenes, torch, mesh, tmp_path, k)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, rss0 = torch.cuda.mem_get_info(0)[0], _rss_mb()
    for k in range(40):
        _cycle(rt, scenes, torch, mesh, tmp_path, k)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, rss1 = torch.cuda.mem_get_info(0)[0], _rss_mb()
    assert free0 - free1 < 32 * 2**20, f"device memory: {(free0 - free1) / 2**20:.1f} MB gone after 40 context lifetimes"
    assert rss1 - rss0 < 64.0, f"host memory: resident set grew by {rss1 - rss0:.1f} MB over 40 context lifetimes"
