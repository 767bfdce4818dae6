"""rt_render_batch_device (include/rtcore.h): several frames of one image shape through ONE pipeline (one cull, one scan and one
compaction launch, one persistent traversal launch over all the frames' pixel lists).  Every frame of a batch must equal
the frame rt_render_device gives for the same uniforms bit for bit — image, hit ids, hit distances — whatever the order
the batch walks the pixels in (hints on / off / converged), for every mode, layout, stripe share and with textures; one
frame is pinned to the oracle as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
COUNT_KEYS = ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests", "hit_updates", "pixels", "stack_drops", "pixels_traced")


def _bufs(torch, n, rows, w):
    dev = torch.device("cuda:0")
    return (torch.full((n, rows, w), 0x01020304, dtype=torch.int32, device=dev),
            torch.full((n, rows, w), 7, dtype=torch.int32, device=dev),
            torch.full((n, rows, w), -1.0, dtype=torch.float32, device=dev))


def _single(ctx, torch, Us, opts, rows=None):
    w, h = int(Us[0][36]), int(Us[0][37])
    rows = h if rows is None else rows
    rgba, tri, t = _bufs(torch, len(Us), rows, w)
    torch.cuda.synchronize()
    for f, U in enumerate(Us):
        ctx.render_device(U, opts, rgba[f].data_ptr(), tri[f].data_ptr(), t[f].data_ptr(), want_stats=False)
    ctx.synchronize()
    return rgba.cpu().numpy(), tri.cpu().numpy(), t.cpu().numpy()


def _batch(ctx, torch, Us, opts, rows=None):
    w, h = int(Us[0][36]), int(Us[0][37])
    rows = h if rows is None else rows
    n = len(Us)
    rgba, tri, t = _bufs(torch, n, rows, w)
    torch.cuda.synchronize()
    ctx.render_batch_device(np.stack(Us), opts, [rgba[f].data_ptr() for f in range(n)], [tri[f].data_ptr() for f in range(n)],
                            [t[f].data_ptr() for f in range(n)])
    ctx.synchronize()
    return rgba.cpu().numpy(), tri.cpu().numpy(), t.cpu().numpy()


def _same(a, b, what):
    for x, y, k in zip(a, b, ("rgba8", "hit_tri", "hit_t")):
        assert np.array_equal(x, y), f"{what}: {k} differs between rt_render_batch_device and per-frame rt_render_device"


def _views(rt, w, h, n, **kw):
    """n frames of a camera orbit (different cull rectangles, different lists) with different time seeds."""
    return [rt.make_uniforms(w, h, cam_pos=(0.25 * np.sin(0.7 * f), 0.1 * f, 1.3 + 0.15 * f), yaw=0.05 * f, time=0.37 * f, **kw)
            for f in range(n)]


def test_batch_equals_per_frame_rendering(rt, oracle):
    import torch
    from raytracer_amd import scenes
    mesh = rt.normalize_mesh(scenes.s70k())
    ctx = rt.Context(0)
    try:
        ctx.set_mesh(mesh)
        ctx.set_surface(mesh)
        ctx.build_lbvh()
        shadow = rt.render_opts(mode=rt.MODE_SHADOW)
        Us = _views(rt, 483, 271, 5)                                     # ragged tiles
        ref = _single(ctx, torch, Us, shadow)
        hit = ref[1].view(np.uint32) != 0xFFFFFFFF
        assert all(0.02 < hit[f].mean() < 0.98 for f in range(5))
        assert len({hit[f].tobytes() for f in range(5)}) == 5            # five different frames
        for rep in range(3):                                             # hints: none, first, converged
            _same(_batch(ctx, torch, Us, shadow), ref, f"shadow, 5 frames, repetition {rep}")
        ctx.set_coherent_scheduling(False)
        _same(_batch(ctx, torch, Us, shadow), ref, "shadow, hints off")
        ctx.set_coherent_scheduling(True)
        # every batch size, the same view 8 times, and a batch after single frames of another shape
        for n in (1, 2, 8, 11, 16, 17, 20, 32):     # 17+: the frame records no longer fit one launch's arguments, the run table one wavefront
            Un = _views(rt, 483, 271, n)
            _same(_batch(ctx, torch, Un, shadow), _single(ctx, torch, Un, shadow), f"shadow, {n} frames")
        same8 = [Us[1]] * 8
        got = _batch(ctx, torch, same8, shadow)
        for f in range(8):
            _same([g[f] for g in got], [r[1] for r in ref], f"8 x the same view, frame {f}")
        small = _views(rt, 96, 64, 3)
        _same(_batch(ctx, torch, small, shadow), _single(ctx, torch, small, shadow), "96x64 after 483x271")
        # the oracle's frame
        f_ = ctx.flatten_for_ssbo()
        exp, _ = oracle.render(f_["bvhData"], f_["triPos"], f_["triNor"], small[2], mode=1, n_threads=8)
        got = _batch(ctx, torch, small, shadow)
        assert np.array_equal(got[1][2].view(np.uint32), exp["hit_tri"]) and np.array_equal(got[2][2], exp["hit_t"])
        assert np.array_equal(got[0][2].view(np.uint8).reshape(64, 96, 4), exp["rgba8"])

        # primary, PT (one lane per pixel in a batch, PT split in rt_render_device: same arithmetic), AOS32
        prim = rt.render_opts(mode=rt.MODE_PRIMARY)
        _same(_batch(ctx, torch, Us, prim), _single(ctx, torch, Us, prim), "primary")
        pt = rt.render_opts(mode=rt.MODE_PT, gi_samples=4, gi_bounces=2)
        Upt = _views(rt, 160, 96, 3)
        _same(_batch(ctx, torch, Upt, pt), _single(ctx, torch, Upt, pt), "PT")
        aos = rt.render_opts(mode=rt.MODE_SHADOW, layout=rt.LAYOUT_AOS32)
        _same(_batch(ctx, torch, Us, aos), ref, "AOS32")
        # stripe shares (packed rows), each rank's share of every frame
        for rank in range(3):
            so = rt.render_opts(mode=rt.MODE_SHADOW, stripe_rows=16, stripe_rank=rank, stripe_world=3)
            rows = rt.packed_rows(271, 16, rank, 3)
            _same(_batch(ctx, torch, Us, so, rows=rows), _single(ctx, torch, Us, so, rows=rows), f"stripes of rank {rank}")
        # textures
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        ctx.set_texture(0, scenes.procedural_texture(256, 0))
        ctx.set_texture(1, scenes.procedural_texture(128, 1))
        tex = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
        rt_ = _single(ctx, torch, Us, tex)
        assert not np.array_equal(rt_[0], ref[0])
        _same(_batch(ctx, torch, Us, tex), rt_, "textured")
        # pipelines in flight: batches rotate over the frame streams (own scratch each); every batch keeps its own frames
        ctx.set_frames_in_flight(2)
        sets = [_bufs(torch, 5, 271, 483) for _ in range(4)]
        Uk = [[Us[(f + k) % 5] for f in range(5)] for k in range(4)]
        for rep in range(2):
            for k in range(4):                                           # four batches queued back to back, two in flight
                ctx.render_batch_device(np.stack(Uk[k]), shadow, [sets[k][0][f].data_ptr() for f in range(5)],
                                        [sets[k][1][f].data_ptr() for f in range(5)], [sets[k][2][f].data_ptr() for f in range(5)])
            ctx.synchronize()
            for k in range(4):
                got = [b.cpu().numpy() for b in sets[k]]
                for f in range(5):
                    _same([g[f] for g in got], [r[(f + k) % 5] for r in ref], f"two batches in flight, batch {k}, frame {f}, round {rep}")
        ctx.set_frames_in_flight(1)
        # outputs are optional per array and per frame
        w, h = 483, 271
        rgba, tri, t = _bufs(torch, 5, h, w)
        ctx.render_batch_device(np.stack(Us), shadow, [rgba[f].data_ptr() if f != 2 else 0 for f in range(5)], None, None)
        ctx.synchronize()
        got = rgba.cpu().numpy()
        for f in range(5):
            assert np.array_equal(got[f], ref[0][f]) if f != 2 else np.all(got[f] == 0x01020304)
        assert np.all(tri.cpu().numpy() == 7)
    finally:
        ctx.close()


def test_traverse_shade_split_is_the_same_frames(rt, monkeypatch):
    """RTCORE_TWO_PHASE=1: RT_MODE_PRIMARY / SHADOW batches as two launches (the primary rays' walk, then the colour of the hits
    compacted per work item).  Not the default any more (a register cap gives the one-launch kernel the same occupancy), kept as
    an option: same images, hit ids, distances and per-frame counters."""
    import torch
    from raytracer_amd import scenes
    mesh = rt.normalize_mesh(scenes.s70k())
    ctx = rt.Context(0)
    try:
        ctx.set_mesh(mesh)
        ctx.set_surface(mesh)
        ctx.build_lbvh()
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        ctx.set_texture(0, scenes.procedural_texture(256, 0))
        ctx.set_texture(1, scenes.procedural_texture(128, 1))
        Us = _views(rt, 483, 271, 6)
        n = len(Us)
        for opts, rows in ((rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED | rt.FLAG_COUNT_WORK), None),
                           (rt.render_opts(mode=rt.MODE_PRIMARY, flags=rt.FLAG_COUNT_WORK), None),
                           (rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_COUNT_WORK, stripe_rows=16, stripe_rank=1, stripe_world=3), rt.packed_rows(271, 16, 1, 3))):
            monkeypatch.setenv("RTCORE_TWO_PHASE", "0")
            one = _batch(ctx, torch, Us, opts, rows=rows)
            st_one = ctx.batch_stats(n)
            monkeypatch.setenv("RTCORE_TWO_PHASE", "1")
            two = _batch(ctx, torch, Us, opts, rows=rows)
            st_two = ctx.batch_stats(n)
            _same(two, one, f"split vs one launch, mode {opts.mode}")
            for f in range(n):
                for k in COUNT_KEYS:
                    assert st_one[f][k] == st_two[f][k], (f, k)
            assert sum(s["rays_primary"] for s in st_one) > 0
    finally:
        monkeypatch.delenv("RTCORE_TWO_PHASE", raising=False)
        ctx.close()


def test_batch_rejects_what_it_does_not_define(rt, plane_mesh):
    ctx = rt.Context(0)
    try:
        o = rt.render_opts(mode=rt.MODE_SHADOW)
        U = rt.make_uniforms(64, 64, **PLANE_VIEW)
        with pytest.raises(rt.RtError) as e:                              # no BVH yet
            ctx.render_batch_device(np.stack([U, U]), o, None, None, None)
        assert e.value.code == -1
        ctx.set_mesh(plane_mesh)
        ctx.build_lbvh()
        ctx.render_batch_device(np.stack([U, U]), o, None, None, None)   # no outputs at all: runs, writes nothing
        ctx.synchronize()
        # a rank that owns no rows of the image (more ranks than stripes): nothing to do, no error
        ctx.render_batch_device(np.stack([U, U]), rt.render_opts(mode=rt.MODE_SHADOW, stripe_rows=64, stripe_rank=1, stripe_world=2),
                                None, None, None)
        ctx.synchronize()
        with pytest.raises(rt.RtError) as e:
            ctx.render_batch_device(np.stack([U] * 33), o, None, None, None)
        assert e.value.code == -1 and "n_frames" in str(e.value)
        with pytest.raises(rt.RtError) as e:
            ctx.render_batch_device(np.stack([U, rt.make_uniforms(64, 72, **PLANE_VIEW)]), o, None, None, None)
        assert e.value.code == -1 and "resolution" in str(e.value)
        with pytest.raises(rt.RtError) as e:                              # counters only of a batch that was asked to count
            ctx.batch_stats(2)
        assert e.value.code == -1
        ctx.render_batch_device(np.stack([U, U]), rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_COUNT_WORK), None, None, None)
        with pytest.raises(rt.RtError) as e:
            ctx.batch_stats(3)                                           # that batch had 2 frames
        assert e.value.code == -1
        assert len(ctx.batch_stats(2)) == 2
        with pytest.raises(rt.RtError) as e:                              # the fragment modes: WIDE64 and untextured only, as for single frames
            ctx.render_batch_device(np.stack([U, U]), rt.render_opts_for_mode(rt.MODE_FRAG_WEB, flags=rt.FLAG_TEXTURED), None, None, None)
        assert e.value.code == -1
        with pytest.raises(rt.RtError) as e:
            ctx.render_batch_device(np.stack([U, U]), rt.render_opts(mode=9), None, None, None)
        assert e.value.code == -1
    finally:
        ctx.close()


def test_batch_counters_equal_the_oracle_per_frame(rt, oracle):
    """RT_FLAG_COUNT_WORK in a batch (rt_batch_stats): every frame of a 20-frame batch carries its own counters, equal to
    the oracle's for that frame's view (and so to rt_render_device's), whatever chunks of which frames a wave happened to
    work on; duplicate views inside a batch (one shared cull / compaction result) count and render like any other frame."""
    import torch
    from raytracer_amd import scenes
    mesh = rt.normalize_mesh(scenes.s70k())
    ctx = rt.Context(0)
    try:
        ctx.set_mesh(mesh)
        ctx.build_lbvh()
        f_ = ctx.flatten_for_ssbo()
        w, h = 243, 135
        views = [rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0)) for k in range(18)]   # the bench's camera path
        Us = views[:7] + [views[3]] + views[7:] + [views[0]]            # 20 frames, two of them duplicates of earlier ones
        assert len(Us) == 20
        for mode, extra in ((rt.MODE_SHADOW, {}), (rt.MODE_PT, dict(gi_samples=2, gi_bounces=2))):
            counted = rt.render_opts(mode=mode, flags=rt.FLAG_COUNT_WORK, **extra)
            plain = rt.render_opts(mode=mode, **extra)
            ref = _single(ctx, torch, Us, plain)
            for rep in range(2):                                         # without and with hints
                _same(_batch(ctx, torch, Us, counted), ref, f"counted batch, mode {mode}")
                st = ctx.batch_stats(20)
                for f in (0, 3, 7, 12, 19) if mode == rt.MODE_PT else range(20):
                    _, ost = oracle.render(f_["bvhData"], f_["triPos"], f_["triNor"], Us[f], mode=mode, n_threads=8, want=(), **extra)
                    for k in COUNT_KEYS[:-1]:
                        assert st[f][k] == ost[k], (mode, rep, f, k, st[f][k], ost[k])
                    one = ctx.render_device(Us[f], counted)
                    assert st[f]["pixels_traced"] == one["pixels_traced"]
                assert st[7] == st[3] and st[19] == st[0]
            _same(_batch(ctx, torch, Us, plain), ref, "duplicate views, uncounted")
        # a stripe share counts its own rows only
        so = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_COUNT_WORK, stripe_rows=16, stripe_rank=1, stripe_world=3)
        rows = rt.packed_rows(h, 16, 1, 3)
        _batch(ctx, torch, Us[:5], so, rows=rows)
        st = ctx.batch_stats(5)
        for f in range(5):
            one = ctx.render_device(Us[f], so)
            for k in COUNT_KEYS:
                assert st[f][k] == one[k], (f, k)
            assert st[f]["pixels"] == rows * w
    finally:
        ctx.close()


@pytest.mark.gpu
def test_near_and_far_record_addresses_are_the_same_frames(rt, monkeypatch):
    """The traversal kernels address their records as scalar base + 32-bit offset while both tables lie in one 4 GB window
    (LAYOUT = RT_LAYOUT_WIDE64_NEAR) and by 64-bit pointer arithmetic otherwise (cfg 5's tables: tests/test_gpu_cfg5_full.py).
    RTCORE_FAR_ADDRESSES=1 selects the 64-bit form on a small scene: same images, hit ids, distances - batches and single frames,
    shadow and PT."""
    import torch
    from raytracer_amd import scenes
    mesh = rt.normalize_mesh(scenes.s70k())
    ctx = rt.Context(0)
    try:
        ctx.set_mesh(mesh)
        ctx.build_lbvh()
        Us = _views(rt, 483, 271, 4)
        for opts in (rt.render_opts(mode=rt.MODE_SHADOW), rt.render_opts(mode=rt.MODE_PT, gi_samples=2, gi_bounces=2)):
            monkeypatch.setenv("RTCORE_FAR_ADDRESSES", "0")
            near = _batch(ctx, torch, Us, opts)
            near_one = _single(ctx, torch, Us, opts)
            monkeypatch.setenv("RTCORE_FAR_ADDRESSES", "1")
            far = _batch(ctx, torch, Us, opts)
            far_one = _single(ctx, torch, Us, opts)
            _same(far, near, f"far vs near addresses, batch, mode {opts.mode}")
            _same(far_one, near_one, f"far vs near addresses, single frames, mode {opts.mode}")
            _same(near, near_one, f"batch vs single frames, mode {opts.mode}")
    finally:
        monkeypatch.delenv("RTCORE_FAR_ADDRESSES", raising=False)
