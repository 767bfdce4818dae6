"""GPU: the windowed exchange of the multi-GPU frame loop (rt_frame_rect, rt_assemble_device_on, dist.BatchedFrames).
rt_frame_rect must be conservative (every pixel outside the rectangle is a miss with the value it reports), rt_assemble_device_on
must rebuild the image rt_deinterleave_device builds from whole stripe buffers, and the batched loop must deliver the same frames
whether it exchanges windows or whole stripes - for views with the model centred, partly off screen, behind the camera (empty
rectangle) and around the camera (rectangle = whole image)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VIEWS = [dict(cam_pos=(0.0, 0.0, 1.3)), dict(cam_pos=(0.9, 0.1, 1.6), yaw=0.15), dict(cam_pos=(0.0, 0.0, 1.3), yaw=1.2),
         dict(cam_pos=(0.0, 0.0, 0.1)), dict(cam_pos=(6.0, 0.0, 1.3)), dict(cam_pos=(-0.4, 0.7, 2.4), pitch=-0.3),
         dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(30.0, 45.0, 10.0), scale=(0.4, 0.4, 0.4))]


@pytest.fixture(scope="module")
def s70k_ctx(rt):
    from raytracer_amd import scenes
    ctx = rt.Context(0)
    ctx.set_mesh(rt.normalize_mesh(scenes.s70k()))
    ctx.build_lbvh()
    yield ctx
    ctx.close()


def test_frame_rect_is_conservative(rt, s70k_ctx):
    w, h = 322, 187
    kinds = set()
    for mode in (rt.MODE_PRIMARY, rt.MODE_SHADOW):
        for v in VIEWS:
            U = rt.make_uniforms(w, h, **v)
            (x0, y0, x1, y1), miss = s70k_ctx.frame_rect(U)
            out, _ = s70k_ctx.render(U, rt.render_opts(mode=mode))
            inside = np.zeros((h, w), bool)
            if x1 >= x0 and y1 >= y0:
                inside[max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = True
            assert np.all(out["hit_tri"][~inside] == 0xFFFFFFFF), v
            assert np.all(out["rgba8"][~inside].view(np.uint32) == miss), v
            kinds.add("empty" if not inside.any() else "whole" if inside.all() else "part")
    assert kinds == {"empty", "whole", "part"}


def test_assemble_equals_deinterleave(rt, s70k_ctx):
    import torch
    w, h, world, sr = 200, 150, 3, 16
    rpr = rt.packed_rows(h, sr, 0, world)
    rng = np.random.default_rng(3)
    packed = torch.from_numpy(rng.integers(0, 2**31, (world, rpr, w), dtype=np.int32)).cuda()
    ref = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    s70k_ctx.deinterleave_device(packed.data_ptr(), ref.data_ptr(), w, h, 4, sr, world, rpr)
    img = torch.full((h, w), 7, dtype=torch.int32, device="cuda:0")
    s70k_ctx.assemble_device(packed.data_ptr(), img.data_ptr(), w, h, sr, world, 0, w, 0, rpr, rpr * w, 0x55)   # the window = everything
    s70k_ctx.synchronize()
    assert torch.equal(img, ref)
    # a real window: columns 40 .. 147, stripe groups 1 .. 2 (image rows 48 .. 143), fill elsewhere
    col0, n_cols, g0, n_rows = 40, 108, 1, 2 * sr
    sub = packed[:, g0 * sr:g0 * sr + n_rows, col0:col0 + n_cols].contiguous()
    s70k_ctx.assemble_device(sub.data_ptr(), img.data_ptr(), w, h, sr, world, col0, n_cols, g0, n_rows, n_rows * n_cols, 0x55)
    s70k_ctx.synchronize()
    exp = np.full((h, w), 0x55, np.int32)
    exp[48:144, col0:col0 + n_cols] = ref.cpu().numpy()[48:144, col0:col0 + n_cols]
    assert np.array_equal(img.cpu().numpy(), exp)
    s70k_ctx.assemble_device(0, img.data_ptr(), w, h, sr, world, 0, 0, 0, 0, 0, 0x66)                          # nothing to place: all fill
    s70k_ctx.synchronize()
    assert np.all(img.cpu().numpy() == 0x66)
    with pytest.raises(rt.RtError):
        s70k_ctx.assemble_device(sub.data_ptr(), img.data_ptr(), w, h, sr, world, 150, 108, 0, n_rows, n_rows * n_cols, 0)   # columns past the image
    with pytest.raises(rt.RtError):
        s70k_ctx.assemble_device(sub.data_ptr(), img.data_ptr(), w, h, sr, world, 0, 108, 0, n_rows, 5, 0)                   # stride smaller than a window


@pytest.fixture(scope="module")
def world1():
    """A one-rank gloo group (device tensors go through host memory, raytracer_amd.dist): the exchange leg forced on."""
    import socket

    import torch.distributed as tdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    yield tdist
    tdist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["root", "all"])
def test_batches_deliver_the_same_frames_with_windows_and_whole_stripes(rt, s70k_ctx, world1, exchange):
    import torch
    from raytracer_amd import dist as rdist
    w, h = 322, 187
    opts = rt.render_opts(mode=rt.MODE_SHADOW)
    Us = [rt.make_uniforms(w, h, **v) for v in VIEWS]
    refs = [s70k_ctx.render(U, opts, want=("rgba8",))[0]["rgba8"] for U in Us]
    dev = torch.device("cuda:0")
    for windowed in (True, False):
        bf = rdist.BatchedFrames(rdist.GpuBackend(s70k_ctx, dev, frames_in_flight=1), w, h, 0, 1, 4, opts, force_collective=True,
                                 exchange=exchange, window_exchange=windowed)
        assert bf.windowed == windowed
        for batch in ((0, 1, 2, 3), (4, 5, 6, 0), (4, 4, 4), (3,), (1, 6)):       # incl. a batch with nothing on screen (4 x 3), and the whole image (3)
            bf.step(np.stack([Us[k] for k in batch]))
            bf.finish()
            for f, k in enumerate(batch):
                got = bf.images[f].cpu().numpy().view(np.uint8).reshape(h, w, 4)
                assert np.array_equal(got, refs[k]), (windowed, batch, f)
        if windowed:
            assert bf.window[0][0][1] < w                 # the last windows were real sub-rectangles
        for fif in (1, 3):                                # the per-frame loop, frames in flight
            fr = rdist.ShardedFrame(rdist.GpuBackend(s70k_ctx, dev, frames_in_flight=fif), w, h, 0, 1, opts, force_collective=True,
                                    exchange=exchange, window_exchange=windowed)
            assert fr.windowed == windowed
            for k in (0, 1, 4, 3, 2, 6, 5, 4, 1):
                fr.step(Us[k], want_stats=False)
            fr.finish()
            assert np.array_equal(fr.image.cpu().numpy().view(np.uint8).reshape(h, w, 4), refs[1]), (windowed, fif)
            st = fr.step(Us[4])                           # nothing on screen, synchronous
            fr.finish()
            assert st["pixels"] == w * h and np.array_equal(fr.image.cpu().numpy().view(np.uint8).reshape(h, w, 4), refs[4])
        s70k_ctx.set_frames_in_flight(1)
