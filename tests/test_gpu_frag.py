"""RT_MODE_FRAG_WEB / RT_MODE_FRAG_ROOT (the older fragment shaders) on the GPU vs the CPU oracle, through
the C ABI.  Same bars as test_gpu_render.py: hit ids bit-exact, t within 1e-5 relative, RGB within 1 LSB — and, since the
kernels are written to be bit-identical, exact equality is asserted as well."""
import numpy as np
import pytest

from test_gpu_render import COUNT_KEYS, PLANE_VIEW, _compare, _scene

pytestmark = pytest.mark.gpu
MODES = {"web": 3, "root": 4}


@pytest.mark.parametrize("mode", ["web", "root"])
def test_plane_frag_modes(gpu_ctx, oracle, rt, plane_mesh, mode):
    ref_b = _scene(gpu_ctx, oracle, plane_mesh)
    w, h = 96, 64
    U = rt.make_uniforms(w, h, **PLANE_VIEW)
    o = rt.render_opts_for_mode(MODES[mode], flags=rt.FLAG_COUNT_WORK)
    assert (o.soft_samples, o.max_stack) == (8, 16) and o.gi_samples == (32 if mode == "web" else 16)
    out, st = gpu_ctx.render(U, o)
    ref, rst = oracle.render(ref_b["nodes"], ref_b["tri_pos"], ref_b["tri_nor"], U, mode=o.mode, gi_samples=o.gi_samples,
                             gi_bounces=o.gi_bounces, soft_samples=o.soft_samples, max_stack=o.max_stack, n_threads=8)
    exact = _compare(out, ref, f"plane/frag-{mode}")
    hit = ref["hit_tri"] != 0xFFFFFFFF
    assert hit.mean() > 0.2
    assert np.all(out["rgba8"][~hit] == np.array([153, 204, 255, 255], np.uint8))         # sky, also on culled tiles
    for k in COUNT_KEYS:
        assert st[k] == rst[k], f"{k}: gpu {st[k]} vs oracle {rst[k]}"
    assert exact == 1.0 and np.array_equal(out["hit_t"], ref["hit_t"])


@pytest.mark.parametrize("mode", ["web", "root"])
def test_frag_off_centre_view_stripes_and_v2(gpu_ctx, oracle, rt, mode):
    """A camera beside the mesh (the mirrored cull rectangle matters), reduced sample counts, v2 records, and the rows of
    one rank of a 3-way stripe split equal the same rows of the full frame."""
    from raytracer_amd import scenes
    mesh = scenes.s70k()
    gpu_ctx.set_node_encoding(rt.NODES_V2)
    try:
        gpu_ctx.set_mesh(mesh)
        gpu_ctx.build_lbvh()
        ref_b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, encoding=2)
        w, h = 200, 120
        U = rt.make_uniforms(w, h, cam_pos=(0.35, 0.6, 1.9), yaw=0.1, pitch=-0.25)
        o = rt.render_opts_for_mode(MODES[mode], gi_samples=5, gi_bounces=2, soft_samples=3)
        out, _ = gpu_ctx.render(U, o)
        ref, _ = oracle.render(ref_b["nodes"], ref_b["tri_pos"], ref_b["tri_nor"], U, mode=o.mode, gi_samples=5, gi_bounces=2,
                               soft_samples=3, max_stack=16, n_threads=8, encoding=2)
        assert _compare(out, ref, f"s70k/frag-{mode}") == 1.0
        hit = ref["hit_tri"] != 0xFFFFFFFF
        assert 0.05 < hit.mean() < 0.95
        rows = hit.any(axis=1).nonzero()[0]
        assert rows.min() > 8 or rows.max() < h - 9          # the mesh is off-centre vertically: a wrong mirror would show
        os_ = rt.render_opts_for_mode(MODES[mode], gi_samples=5, gi_bounces=2, soft_samples=3, stripe_rows=16, stripe_rank=1, stripe_world=3)
        part, _ = gpu_ctx.render(U, os_)                      # the wrapper puts the rank's rows back at their frame positions
        mine = [y for y in range(h) if (y // 16) % 3 == 1]
        for k in ("rgba8", "hit_tri", "hit_t"):
            assert np.array_equal(part[k][mine], out[k][mine]), k
    finally:
        gpu_ctx.set_node_encoding(rt.NODES_V1)


def test_frag_argument_errors(gpu_ctx, rt, plane_mesh):
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    U = rt.make_uniforms(32, 32, **PLANE_VIEW)
    for bad in (dict(soft_samples=9), dict(layout=rt.LAYOUT_AOS32), dict(flags=rt.FLAG_TEXTURED)):
        with pytest.raises(rt.RtError):
            gpu_ctx.render(U, rt.render_opts_for_mode(rt.MODE_FRAG_WEB, **bad))
    with pytest.raises(rt.RtError):
        gpu_ctx.render(U, rt.render_opts_for_mode(rt.MODE_FRAG_ROOT, gi_samples=17))


@pytest.mark.parametrize("mode", ["web", "root"])
def test_frag_batches_equal_single_frames(gpu_ctx, rt, plane_mesh, mode):
    """rt_render_batch_device with the fragment-shader modes: frame f of a batch = rt_render_device of its uniforms, bit for bit
    (image, hit ids, distances) and counter for counter; views differ (the mirrored cull rectangle is per frame)."""
    import torch
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    w, h, n = 160, 96, 5
    Us = np.stack([rt.make_uniforms(w, h, cam_pos=(0.15 * k - 0.3, 0.2 * k - 0.4, 1.3 + 0.1 * k), rot_deg=(-90.0, 7.0 * k, 0.0),
                                    scale=(2.5, 2.5, 2.5)) for k in range(n)])
    o = rt.render_opts_for_mode(MODES[mode], gi_samples=4, gi_bounces=2, soft_samples=3, flags=rt.FLAG_COUNT_WORK)
    dev = torch.device("cuda:0")
    buf = {k: torch.zeros((n, h, w), dtype=dt, device=dev) for k, dt in (("rgba8", torch.int32), ("hit_tri", torch.int32), ("hit_t", torch.float32))}
    gpu_ctx.render_batch_device(Us, o, *[[buf[k][f].data_ptr() for f in range(n)] for k in ("rgba8", "hit_tri", "hit_t")])
    stats = gpu_ctx.batch_stats(n)
    hits = 0
    for f in range(n):
        ref, st = gpu_ctx.render(Us[f], o)
        assert np.array_equal(buf["rgba8"][f].cpu().numpy().view(np.uint8).reshape(h, w, 4), ref["rgba8"]), f
        assert np.array_equal(buf["hit_tri"][f].cpu().numpy().view(np.uint32), ref["hit_tri"]), f
        assert np.array_equal(buf["hit_t"][f].cpu().numpy(), ref["hit_t"]), f
        for k in COUNT_KEYS:
            assert stats[f][k] == st[k], (f, k)
        hits += int((ref["hit_tri"] != 0xFFFFFFFF).sum())
    assert hits > 1000
    with pytest.raises(rt.RtError):
        gpu_ctx.render_batch_device(Us, rt.render_opts_for_mode(MODES[mode], layout=rt.LAYOUT_AOS32), [buf["rgba8"][f].data_ptr() for f in range(n)])
