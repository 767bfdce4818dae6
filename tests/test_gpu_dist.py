"""The multi-GPU frame loop's stream plumbing on ONE GPU: a world_size-1 RCCL group with the
gather + de-interleave leg forced on, frames queued without host synchronisation (the path
bench.py times), against the synchronous single-call frame.  The N>1 host logic itself
(stripe ownership, packed layout, gather order) is covered on CPU by tests/test_dist_cpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))


@pytest.fixture(scope="module")
def nccl_world1():
    import os
    import socket

    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all", "root"])
@pytest.mark.parametrize("fif", [1, 2, 4])
def test_async_frames_with_forced_collective(gpu_ctx, rt, plane_mesh, nccl_world1, fif, exchange):
    import torch

    from raytracer_amd import dist as rdist
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    w, h = 320, 200
    opts = rt.render_opts(mode=rt.MODE_SHADOW)
    dev = torch.device("cuda:0")
    frame = rdist.ShardedFrame(rdist.GpuBackend(gpu_ctx, dev, frames_in_flight=fif), w, h, 0, 1, opts, force_collective=True,
                               exchange=exchange)
    views = [dict(PLANE_VIEW), dict(PLANE_VIEW, cam_pos=(0.1, 0.0, 1.2)), dict(PLANE_VIEW, rot_deg=(-70.0, 10.0, 0.0))]
    gpu_ctx.profile_begin()
    for k in range(9):                      # queued back to back: render k+1 must not overwrite stripes still being gathered
        frame.step(rt.make_uniforms(w, h, **views[k % 3]), want_stats=False)
    frame.finish()
    ms_frame, ms_trav, n = gpu_ctx.profile_end()
    assert n == 9 and 0.0 < ms_trav <= ms_frame
    got = frame.image.cpu().numpy().view(np.uint8).reshape(h, w, 4)
    ref, _ = gpu_ctx.render(rt.make_uniforms(w, h, **views[8 % 3]), opts, want=("rgba8",))
    assert np.array_equal(got, ref["rgba8"])
    assert (ref["rgba8"][..., :3].max(axis=2) > 0).mean() > 0.05
    gpu_ctx.set_frames_in_flight(1)


@pytest.mark.parametrize("exchange", ["all", "root"])
@pytest.mark.parametrize("fif", [1, 2])
def test_batched_frames_with_forced_collective(gpu_ctx, rt, plane_mesh, nccl_world1, exchange, fif):
    """dist.BatchedFrames (the loop bench.py runs): 3 frames per pipeline, the batch's ONE collective and its de-interleaves on
    a side stream behind an event while the context stream already traverses the next batch; two buffer sets alternate.
    Every frame must be the frame of its own uniforms, including a short last batch."""
    import torch

    from raytracer_amd import dist as rdist
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    w, h = 320, 200
    opts = rt.render_opts(mode=rt.MODE_SHADOW)
    dev = torch.device("cuda:0")
    views = [dict(PLANE_VIEW), dict(PLANE_VIEW, cam_pos=(0.1, 0.0, 1.2)), dict(PLANE_VIEW, rot_deg=(-70.0, 10.0, 0.0)),
             dict(PLANE_VIEW, cam_pos=(-0.2, 0.1, 1.0)), dict(PLANE_VIEW, rot_deg=(-80.0, -15.0, 5.0))]
    Us = [rt.make_uniforms(w, h, **v) for v in views]
    refs = [gpu_ctx.render(U, opts, want=("rgba8",))[0]["rgba8"] for U in Us]
    assert len({r.tobytes() for r in refs}) == 5
    bf = rdist.BatchedFrames(rdist.GpuBackend(gpu_ctx, dev, frames_in_flight=fif), w, h, 0, 1, 3, opts, force_collective=True,
                             exchange=exchange)
    img = lambda f: bf.images[f].cpu().numpy().view(np.uint8).reshape(h, w, 4)
    for rounds in range(3):                     # 7 batches queued back to back per round: buffer sets are reused 3 times
        order = [(0, 1, 2), (3, 4, 0), (1, 2, 3), (4, 0, 1), (2, 3, 4), (0, 2, 4), (1, 3)]
        for b in order:
            bf.step(np.stack([Us[k] for k in b]))
        bf.finish()
        assert np.array_equal(img(0), refs[1]) and np.array_equal(img(1), refs[3])       # the short last batch
        assert np.array_equal(img(2), refs[4])                                            # left over from the batch before it
    # one full batch alone, and the frames after a per-frame call in between
    bf.step(np.stack([Us[4], Us[3], Us[2]]))
    bf.finish()
    assert all(np.array_equal(img(f), refs[k]) for f, k in enumerate((4, 3, 2)))
    out, _ = gpu_ctx.render(Us[0], opts, want=("rgba8",))
    assert np.array_equal(out["rgba8"], refs[0])
    bf.step(np.stack([Us[0], Us[1], Us[2]]))
    bf.finish()
    assert all(np.array_equal(img(f), refs[k]) for f, k in enumerate((0, 1, 2)))
    gpu_ctx.set_frames_in_flight(1)


def test_two_frames_in_flight_without_collective(gpu_ctx, rt, plane_mesh):
    """rt_set_frames_in_flight(2): asynchronous frames alternate between two frame streams and two output buffers; every
    frame must still be the frame of its own uniforms, and synchronous calls in between see finished results."""
    import torch

    from raytracer_amd import dist as rdist
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    w, h = 400, 240
    opts = rt.render_opts(mode=rt.MODE_SHADOW)
    dev = torch.device("cuda:0")
    views = [dict(PLANE_VIEW), dict(PLANE_VIEW, cam_pos=(0.1, 0.0, 1.2)), dict(PLANE_VIEW, rot_deg=(-70.0, 10.0, 0.0)),
             dict(PLANE_VIEW, cam_pos=(-0.2, 0.1, 1.0))]
    refs = [gpu_ctx.render(rt.make_uniforms(w, h, **v), opts, want=("rgba8",))[0]["rgba8"] for v in views]
    assert not np.array_equal(refs[0], refs[1])
    frame = rdist.ShardedFrame(rdist.GpuBackend(gpu_ctx, dev, frames_in_flight=3), w, h, 0, 1, opts)
    for rounds in range(3):
        imgs = []
        for k in range(8):
            frame.step(rt.make_uniforms(w, h, **views[k % 4]), want_stats=False)
            imgs.append(frame.image)                      # two alternating buffers: keep the last two
        frame.finish()
        a = imgs[-1].cpu().numpy().view(np.uint8).reshape(h, w, 4)
        b = imgs[-2].cpu().numpy().view(np.uint8).reshape(h, w, 4)
        assert np.array_equal(a, refs[7 % 4]) and np.array_equal(b, refs[6 % 4])
        # a synchronous frame (stats) right behind asynchronous ones joins them first
        frame.step(rt.make_uniforms(w, h, **views[1]), want_stats=False)
        out, st = gpu_ctx.render(rt.make_uniforms(w, h, **views[2]), opts, want=("rgba8",))
        assert np.array_equal(out["rgba8"], refs[2]) and st["pixels"] == w * h
        frame.finish()
    gpu_ctx.set_frames_in_flight(1)


def test_profile_end_needs_begin(gpu_ctx, rt):
    with pytest.raises(rt.RtError):
        gpu_ctx.profile_end()


def test_caller_buffers_with_4_byte_alignment(gpu_ctx, rt, plane_mesh):
    """rt_render_device takes any 4-byte-aligned device pointers: k_cull writes whole culled rows with 16-byte stores only
    where the pointer allows it.  Camera A (almost every pixel culled) into buffers offset by 4, 8 and 12 bytes."""
    import torch
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    w, h = 200, 120
    opts = rt.render_opts(mode=rt.MODE_SHADOW)
    for view in (dict(cam_pos=(0.0, 0.0, 5.0)), dict(PLANE_VIEW)):
        U = rt.make_uniforms(w, h, **view)
        ref, _ = gpu_ctx.render(U, opts)
        for off in (1, 2, 3):
            bufs = [torch.full((w * h + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0") for _ in range(3)]
            ptr = [b.data_ptr() + 4 * off for b in bufs]
            gpu_ctx.render_device(U, opts, d_rgba8=ptr[0], d_hit_tri=ptr[1], d_hit_t=ptr[2])
            got = [b.cpu().numpy() for b in bufs]
            for g in got:                                   # nothing outside [off, off + w h) was touched
                assert np.all(g[:off] == 0x5A5A5A5A) and np.all(g[off + w * h:] == 0x5A5A5A5A)
            assert np.array_equal(got[0][off:off + w * h].view(np.uint8).reshape(h, w, 4), ref["rgba8"])
            assert np.array_equal(got[1][off:off + w * h].view(np.uint32).reshape(h, w), ref["hit_tri"])
            assert np.array_equal(got[2][off:off + w * h].view(np.float32).reshape(h, w), ref["hit_t"])
    with pytest.raises(rt.RtError):
        gpu_ctx.render_device(U, opts, d_rgba8=bufs[0].data_ptr() + 2)
