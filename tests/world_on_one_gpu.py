#!/usr/bin/env python3
"""One rank of an N-rank rehearsal of raytracer_amd.dist on a ONE-GPU box (tests/test_gpu_world.py launches N of these with
torch.distributed.run): every rank has its own rt context on cuda:0, the collectives run over gloo, staged through host memory
by raytracer_amd.dist (RCCL refuses two ranks on one device; RT_WORLD_VERBOSE=1 prints the stages).  Everything else is the real N > 1 path: rank 0 builds, replicate_bvh (scatter + all-gather of
one blob, rt_bvh_adopt_device / rt_adopt_tri_uv_device on the other ranks), every rank renders ITS stripes, the packed stripes
are exchanged ("all" / "root"), rt_deinterleave_device[_on] assembles the frames.  Checked on every rank that holds a frame:
the assembled frame equals the rank's own render of the whole frame, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import raytracer_amd as rt
from raytracer_amd import dist as rdist, scenes


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)     # a rank stuck in a collective says where, and ends
    say = (lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True)) if os.environ.get("RT_WORLD_VERBOSE") else (lambda m: None)
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    W, H = 483, 271                                       # 17 stripes of 16 rows, the last one 15 rows: ragged for every N
    for enc in (rt.NODES_V1, rt.NODES_V2):
        ctx = rt.Context(0)
        if rank == 0:
            mesh = rt.normalize_mesh(scenes.s70k())
            ctx.set_node_encoding(enc)
            ctx.set_mesh(mesh)
            ctx.set_surface(mesh)
            ctx.build_lbvh()
        say(f"enc {enc}: built")
        keep = rdist.replicate_bvh(ctx, rank, world, dev)
        say("replicated")
        assert (len(keep) == 0) == (rank == 0)
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        ctx.set_texture(0, scenes.procedural_texture(256, 0))
        ctx.set_texture(1, scenes.procedural_texture(128, 1))
        # the adopted tree is rank 0's, byte for byte
        f = ctx.flatten_for_ssbo()
        digest = torch.tensor([int(np.frombuffer(f[k].tobytes(), np.uint32).astype(np.uint64).sum() % (1 << 62)) for k in ("bvhData", "triPos", "triNor")]
                              + [int(ctx.tri_uv().view(np.uint32).astype(np.uint64).sum()), ctx.node_encoding()], dtype=torch.int64)
        every = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(every, digest)
        assert all(torch.equal(e, every[0]) for e in every), f"rank {rank}: adopted BVH differs from rank 0's"
        assert ctx.node_encoding() == enc

        opts = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
        views = [rt.make_uniforms(W, H, cam_pos=(0.25 * np.sin(0.7 * k), 0.1 * k, 1.3 + 0.15 * k), yaw=0.05 * k, time=0.37 * k) for k in range(7)]
        refs = [ctx.render(U, opts, want=("rgba8",))[0]["rgba8"] for U in views]          # this rank's own whole frames
        lit = [float((r[..., :3].max(axis=2) > 0).mean()) for r in refs]
        assert len({r.tobytes() for r in refs}) == 7 and min(lit) > 0.004, lit
        as_img = lambda t: t.cpu().numpy().view(np.uint8).reshape(H, W, 4)

        for exchange in ("all", "root"):
            holds = rank == 0 or exchange == "all"
            for fif in (1, 3):                            # one frame per pipeline
                say(f"enc {enc} exchange {exchange}: per-frame loop, {fif} in flight")
                fr = rdist.ShardedFrame(rdist.GpuBackend(ctx, dev, frames_in_flight=fif), W, H, rank, world, opts, exchange=exchange)
                assert fr.rpr == rt.packed_rows(H, 16, 0, world) and fr.opts.stripe_world == world
                for k in range(9):
                    fr.step(views[k % 7], want_stats=False)
                fr.finish()
                assert (fr.image is not None) == holds
                if holds:
                    assert np.array_equal(as_img(fr.image), refs[8 % 7]), f"rank {rank}: per-frame loop, exchange {exchange}, fif {fif}"
                st = fr.step(views[3])                    # synchronous, with this rank's share of the work counters
                fr.finish()
                if holds:
                    assert np.array_equal(as_img(fr.image), refs[3])
                mine = len(rdist.stripe_rows_of(H, rank, world))
                assert st["pixels"] == mine * W, (st["pixels"], mine * W)
            for fif in (1, 2):                            # batches of up to 5 frames, the last one short
                say(f"enc {enc} exchange {exchange}: batches, {fif} in flight")
                bf = rdist.BatchedFrames(rdist.GpuBackend(ctx, dev, frames_in_flight=fif), W, H, rank, world, 5, opts, exchange=exchange)
                for rounds in range(2):
                    for b in ((0, 1, 2, 3, 4), (5, 6, 0, 1, 2), (3, 4, 5, 6, 0), (6, 2)):
                        bf.step(np.stack([views[k] for k in b]))
                    bf.finish()
                    assert (bf.images is not None) == holds
                    if holds:
                        for f_, k in enumerate((6, 2, 5, 6, 0)):      # the short batch, then what the batch before it left
                            assert np.array_equal(as_img(bf.images[f_]), refs[k]), f"rank {rank}: batch, exchange {exchange}, fif {fif}, frame {f_}"
            ctx.set_frames_in_flight(1)
        dist.barrier()
        ctx.close()
        del keep
    faulthandler.cancel_dump_traceback_later()
    print(f"OK rank {rank} of {world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
