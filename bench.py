#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: Mrays/s, primary + shadow rays, dragon-class
mesh at 3840x2160 (cfg 3), plus LBVH build ms.  `web/dragon.obj` is a Git-LFS pointer in the
reference, so the workload is the closed-form stand-in S1M (1 310 720 triangles).

  python bench.py --gpus N --steps K --warmup W
  N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one frame(): the traversal kernel over the whole 4K image (inputs resident in HBM),
for N > 1 sharded as 16-row image-space stripes + RCCL all-gather of the packed stripes +
de-interleave, all inside the timed region.  The BVH is built once on rank 0 and replicated
with RCCL broadcasts before the timed region.  Total work is fixed as N grows => "strong".

One JSON line on rank 0; `roofline` is the traversal kernel against the HBM roof using the
ALGORITHMIC bytes of SURVEY §8d (32 V + 48 K + 48 H + 4 P from the kernel's own work counters,
which tests/test_gpu_render.py proves equal to the oracle's); `cpu_baseline` is the scalar
oracle timed on the host cores on a bounded sample of the same frame (reported, not a target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def scene_mesh(name):
    import raytracer_amd as rt
    from raytracer_amd import scenes
    if name == "s1m":
        m = scenes.s1m()
    elif name == "s70k":
        m = scenes.s70k()
    elif name.startswith("lattice"):
        # cfg 5 of SURVEY §8d: K copies of the (normalised) cfg-3 mesh on a lattice, flattened, normalised again
        m = scenes.lattice(rt.normalize_mesh(scenes.s1m()), int(name[len("lattice"):]))
    elif name.startswith("obj:"):
        # a mesh file supplied out of band (bunny.obj / dragon.obj of BASELINE.md are not in the reference's tree): its MTL and
        # the textures the MTL names are looked up beside it
        path = name[4:]
        m = rt.load_obj(path, os.path.dirname(os.path.abspath(path)))
    else:
        raise SystemExit(f"unknown scene {name}")
    return rt.normalize_mesh(m)          # script.js:720: every mesh is normalised into the unit cube


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="s1m", help="s1m | s70k | lattice<K> (closed-form stand-ins, SURVEY §8d) | obj:<file.obj> (a mesh supplied "
                    "out of band; MTL and textures are looked up beside it, JPEG / PNG through PIL)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", default="shadow", choices=["primary", "shadow", "pt", "frag-web", "frag-root"],
                    help="frag-web / frag-root: the older fragment shaders (web/shader.wgsl, shader.wgsl; SURVEY §8f-4) with the "
                         "constants they are written with, constant albedo")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--layout", default="wide64", choices=["wide64", "aos32"])
    ap.add_argument("--encoding", default="auto", choices=["auto", "v1", "v2"],
                    help="node records: v1 = reference (f32-valued indices, < 2^24 nodes), v2 = u32 bit-casts")
    ap.add_argument("--frames-in-flight", default="auto", choices=["auto", "1", "2", "3", "4"],
                    help="asynchronous frames in flight per GPU (rtcore.h rt_set_frames_in_flight). auto = 1 on one GPU, so that every "
                         "k_render launch runs alone and its duration is the kernel's own (roofline), 3 with N > 1, where a rank's "
                         "share of the frame is small and mostly tail (3 frame streams + the context stream = the 4 hardware queues)")
    ap.add_argument("--batch", default="auto", choices=["auto"] + [str(k) for k in range(1, 33)],
                    help="frames per pipeline (rtcore.h rt_render_batch_device). auto = 32 (RT_MAX_BATCH) on one GPU, 16 with N > 1: the K steps run as "
                         "ceil(K / batch) pipelines of equal size (K = 20: one pipeline of 20; N > 1: 10 + 10), each ONE k_render launch over the pixel lists of its frames (a frame, and even more a rank's "
                         "stripe share of one, is mostly launch overhead + idle tail on its own: tools/batch_gain.py); with N > 1 one "
                         "exchange per batch on a side stream.  1 = one frame per pipeline (the `per_frame` leg at N = 1)")
    ap.add_argument("--exchange", default="root", choices=["root", "all"],
                    help="N > 1: where the frame's stripes are gathered — root = rank 0 only (the viewer's rank: grouped send/recv, each "
                         "byte crosses one xGMI link once), all = every rank (all_gather, N times the traffic)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank uses cuda:0 and the collectives run over gloo, staged through host memory (RCCL "
                         "refuses two ranks on one device).  Exercises the whole N > 1 code path (replication, stripes, batched exchange, "
                         "de-interleave, reductions); its numbers mean nothing and the line says so")
    ap.add_argument("--no-fif2", action="store_true", help="skip the extra two-frames-in-flight loop at N = 1")
    ap.add_argument("--no-legs", action="store_true", help="skip the unhinted and moving-camera legs at N = 1")
    ap.add_argument("--no-texture", action="store_true",
                    help="shade with the reference's constant albedo (compute.wgsl:372) instead of the materials + textures of cfg 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)

    mode = {"primary": rt.MODE_PRIMARY, "shadow": rt.MODE_SHADOW, "pt": rt.MODE_PT, "frag-web": rt.MODE_FRAG_WEB,
            "frag-root": rt.MODE_FRAG_ROOT}[args.mode]
    frag = mode in (rt.MODE_FRAG_WEB, rt.MODE_FRAG_ROOT)
    layout = dict(wide64=rt.LAYOUT_WIDE64, aos32=rt.LAYOUT_AOS32)[args.layout]
    W, H = args.width, args.height
    ctx = rt.Context(local)
    from_file = args.scene.startswith("obj:")
    file_mesh = scene_mesh(args.scene) if from_file else None      # every rank reads the file (materials + textures are host-side data)
    textured = not args.no_texture and not frag and (not from_file or (file_mesh.texcoords is not None and file_mesh.texcoords.size > 0
                                                                          and file_mesh.tex_idx is not None and len(file_mesh.materials) > 0))
    tflag = rt.FLAG_TEXTURED if textured else 0

    # ---- hot path 1: build once on rank 0 (timed separately), replicate with RCCL
    build = None
    mesh = None
    encoding = rt.NODES_V1
    if rank == 0:
        mesh = file_mesh if from_file else scene_mesh(args.scene)
        if args.encoding == "v2" or (args.encoding == "auto" and 2 * mesh.n_tri - 1 > (1 << 24)):
            encoding = rt.NODES_V2
        ctx.set_node_encoding(encoding)
        ctx.set_mesh(mesh)
        if textured:
            ctx.set_surface(mesh)       # texcoords + material ids: reordered into BVH order by every build below
        runs, walls = [], []
        for _ in range(5):
            t_b = time.perf_counter()
            runs.append(ctx.build_lbvh())          # returns after the build's one read-back: the tree is usable
            walls.append((time.perf_counter() - t_b) * 1e3)
        runs.sort(key=lambda s: s["ms_total"])
        build = runs[len(runs) // 2]
        build_wall_ms = sorted(walls)[len(walls) // 2]
    bcast_ms = 0.0
    keep = None
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = rdist.replicate_bvh(ctx, rank, world, device)
        dist.barrier()
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3

    bound = None
    if textured and from_file:          # the file's own materials and the images its MTL names (those that exist beside it);
        class _SurfaceDone:             # the surface records themselves were set before the build (rank 0) / came with the BVH
            def __init__(self, c): self.c = c
            def set_surface(self, m): pass
            def __getattr__(self, k): return getattr(self.c, k)
        bound = rt.bind_obj_surface(_SurfaceDone(ctx), file_mesh, os.path.dirname(os.path.abspath(args.scene[4:])))
    elif textured:                      # materials + closed-form textures: small host-side descriptions, set on every rank
        from raytracer_amd import scenes
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        ctx.set_texture(0, scenes.procedural_texture(2048, 0))
        ctx.set_texture(1, scenes.procedural_texture(1024, 1))

    # camera (B) of SURVEY §8d: pos (0,0,1.3), default sliders
    U = rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3))
    shade = rt.render_opts_for_mode(mode)      # sample counts / stack depth the shader file is written with
    if not frag:
        shade.gi_samples = args.spp
    shade = dict(gi_samples=shade.gi_samples, gi_bounces=shade.gi_bounces, soft_samples=shade.soft_samples, max_stack=shade.max_stack)
    base = rt.render_opts(mode=mode, layout=layout, flags=tflag, **shade)
    # auto: 32 (RT_MAX_BATCH) on one GPU - K = 20 steps are ONE pipeline; 16 with N > 1, where two pipelines (10 + 10) let the
    # exchange of the first overlap the traversal of the second; the older fragment shaders run one frame per pipeline
    batch = (1 if frag else (32 if world == 1 else 16)) if args.batch == "auto" else int(args.batch)
    if frag and batch > 1:
        raise SystemExit("--batch > 1 renders through rt_render_batch_device: --mode primary|shadow|pt only")
    # pipelines (frames, or batches of frames) in flight: 1 at N = 1, so that every k_render launch runs alone and its duration is
    # the kernel's own.  Batches stay at 1 with N > 1 too: a second batch in flight gains 5 % on a 1/8 share of a 4K frame and
    # loses 27 % on an 8K frame (each pipeline then owns 2 of the 3 resident blocks per CU); single frames use three
    fif = (1 if (world == 1 or batch > 1) else 3) if args.frames_in_flight == "auto" else int(args.frames_in_flight)
    frame = rdist.ShardedFrame(rdist.GpuBackend(ctx, device, frames_in_flight=fif), W, H, rank, world, base, exchange=args.exchange)
    bframes = rdist.BatchedFrames(frame.b, W, H, rank, world, batch, base, exchange=args.exchange) if batch > 1 else None

    def run_frames(views, n):
        """n steps (frames) of the frame loop, views[i % len(views)] each: one frame per pipeline, or `batch` frames per pipeline."""
        if bframes is None:
            for i in range(n):
                frame.step(views[i % len(views)], want_stats=False)
            return
        i0 = 0
        for size in rdist.batch_sizes(n, batch):             # pipelines of equal size (K = 20, 16 at most per pipeline: 10 + 10)
            bframes.step(np.stack([views[i % len(views)] for i in range(i0, i0 + size)]))
            i0 += size

    # one counted launch (untimed): rays + work counters of this rank's share of the frame
    copts = rt.render_opts(mode=mode, layout=layout, flags=rt.FLAG_COUNT_WORK | tflag, **shade,
                           stripe_rows=frame.opts.stripe_rows, stripe_rank=rank, stripe_world=world)
    def counted(Uc):
        """Rays and algorithmic bytes of one frame (this rank's share), from the kernels' own work counters.
        -> (rays as the reference's loop casts them, rays that enter the tree, bytes of the whole frame, k_render's own bytes).
        Pixels whose primary ray misses the root box never reach k_render (k_cull disposes of them, most of them through a
        screen rectangle): k_stats_final counts each as one ray + one box test + one pixel, and those 36 B are NOT k_render's."""
        c = ctx.render_device(Uc, copts, d_rgba8=frame.packed.data_ptr())
        culled = c["pixels"] - c["pixels_traced"]
        rays = c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
        frame_bytes = 32 * c["box_tests"] + 48 * c["tri_tests"] + 48 * c["hit_updates"] + 4 * c["pixels"]
        return rays, rays - culled, frame_bytes, frame_bytes - 36 * culled

    my_rays, my_rays_traced, my_frame_bytes, my_bytes = counted(U)

    if bframes is not None:             # set-up, like the counted launch above: the first full batch sizes the pipeline's scratch
        bframes.step(np.stack([U] * batch))
        bframes.finish()
    run_frames([U], args.warmup)

    def fence():
        frame.finish()
        if bframes is not None:
            bframes.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    # frames are queued asynchronously, as a frame loop would; hipEvents recorded on the context's own stream
    # around every frame's kernels (rt_profile_begin/end) give the per-kernel durations of exactly these K steps
    ctx.profile_begin()
    run_frames([U], args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, trav_ms, n_launch = ctx.profile_end()   # frame pipeline (cull, compact, traversal, stats) / k_render alone, per launch
    frames_per_launch = args.steps / max(n_launch, 1)  # 1, or the mean batch size (a k_render launch then walks that many frames' lists)

    # N > 1: the frame rank 0 assembled from all ranks' stripes (the last timed one) against rank 0's own render of the whole
    # frame — outside the timed region; says on the driver's 8-GPU run that the exchange delivered the right pixels
    frame_check = None
    if world > 1 and rank == 0:
        try:
            assembled = bframes.images[0] if bframes is not None else frame.image
            whole = torch.zeros((H, W), dtype=torch.int32, device=device)
            ctx.render_device(U, base, d_rgba8=whole.data_ptr(), want_stats=False)
            ctx.synchronize()
            frame_check = "identical" if torch.equal(whole, assembled) else f"{int((whole != assembled).sum())} pixels differ"
        except Exception as e:          # (a check outside the timed region must not cost the line)
            frame_check = f"check failed: {type(e).__name__}: {e}"

    # the same K frames again with two frames in flight (only where the main loop ran with one): whole-job throughput when
    # the tail of one frame may overlap the head of the next; reported beside `value`, never as `value` at N = 1
    fif2 = None
    if fif == 1 and world == 1 and not args.no_fif2 and (batch == 1 or len(rdist.batch_sizes(args.steps, batch)) > 1):   # (one pipeline: nothing to overlap)
        b2 = rdist.GpuBackend(ctx, device, frames_in_flight=2)
        if batch > 1:                   # two BATCHES in flight: cull / compaction and the head of batch i+1 beside the tail of batch i
            frame2 = rdist.BatchedFrames(b2, W, H, rank, world, batch, base)
            go = lambda n: [frame2.step(np.stack([U] * size)) for size in rdist.batch_sizes(n, batch)]
            go(2 * batch)               # both frame slots size their scratch
        else:
            frame2 = rdist.ShardedFrame(b2, W, H, rank, world, base)
            go = lambda n: [frame2.step(U, want_stats=False) for _ in range(n)]
        go(args.warmup)
        frame2.finish()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        go(args.steps)
        frame2.finish()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        fif2 = {"value": round(my_rays * args.steps / e2 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(e2 / args.steps * 1e3, 4),
                "what": "the same K steps with two pipelines in flight (overlapped launches stretch each other, so their durations no "
                        "longer measure the kernel: reported beside `value`, never as `value`)"}
        ctx.set_frames_in_flight(1)

    # Two more legs at N = 1, same frame loop, reported beside `value` (never as `value`): the same static view WITHOUT the
    # frame-coherent scheduling hints, and a view that changes every frame (the model turning 1 degree per frame, as a user
    # dragging the reference's rotY slider), where last frame's per-pixel hints are a few pixels off.
    legs = {}
    if world == 1 and not args.no_legs:
        def leg(uniforms_list, rays_list, bytes_list, per_frame=False):
            go = (lambda n: [frame.step(uniforms_list[i % len(uniforms_list)], want_stats=False) for i in range(n)]) if per_frame \
                else (lambda n: run_frames(uniforms_list, n))
            go(max(args.warmup, batch))
            fence()
            t1 = time.perf_counter()
            ctx.profile_begin()
            go(args.steps)
            fence()
            e = time.perf_counter() - t1
            _, tr_ms, launches = ctx.profile_end()
            n = args.steps
            rays = sum(rays_list[i % len(rays_list)] for i in range(n))
            byts = sum(bytes_list[i % len(bytes_list)] for i in range(n))
            ach = byts / (tr_ms * launches * 1e-3) / 1e9 if tr_ms > 0 else 0.0      # all launches' bytes / all launches' time
            return {"value": round(rays / e / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(e / n * 1e3, 4),
                    "kernel_ms": round(tr_ms, 4), "frames_per_launch": round(n / max(launches, 1), 3), "roofline_frac": round(ach / HBM_PEAK_GBS, 5)}
        if batch > 1:
            legs["per_frame"] = leg([U], [my_rays], [my_bytes], per_frame=True)
            legs["per_frame"]["what"] = "same static view, ONE frame per pipeline (rt_render_device, one frame in flight): the latency-bound loop of an interactive viewer"
        ctx.set_coherent_scheduling(False)
        legs["unhinted"] = leg([U], [my_rays], [my_bytes])
        legs["unhinted"]["what"] = "same static view, frame-coherent scheduling hints off (a first frame / camera cut runs at this speed)"
        ctx.set_coherent_scheduling(True)
        views = [rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0)) for k in range(max(args.steps, 8))]
        cnt = [counted(v) for v in views]
        legs["moving_camera"] = leg(views, [c[0] for c in cnt], [c[3] for c in cnt])
        legs["moving_camera"]["what"] = ("rotY + 1 degree per frame; a batch whose views differ is not ordered by hints (they would be a batch old)"
                                         if batch > 1 else "rotY + 1 degree per frame, hints on (tile-level promotion under motion)")
        run_frames([U], 3 * batch)      # leave the context as the main loop had it: hints of the static view
        fence()

    tot = torch.tensor([float(my_rays), float(my_bytes), elapsed, kernel_ms, float(my_rays_traced)], dtype=torch.float64,
                       device="cpu" if args.rehearse_on_one_gpu else device)     # (gloo rehearsal: reductions on host tensors)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, kernel_max = float(mx[2]), float(mx[3])
    else:
        kernel_max = kernel_ms
    total_rays, total_bytes, total_traced = float(tot[0]), float(tot[1]), float(tot[4])

    if rank == 0:
        pmc = pmc_traffic(args, world, textured, batch, frames_per_launch)
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (traversal): this rank's algorithmic bytes / its average launch duration
        achieved = my_bytes * frames_per_launch / (trav_ms * 1e-3) / 1e9 if trav_ms > 0 else 0.0
        what = {"primary": "primary rays only", "shadow": "primary + 4 soft shadow rays",
                "pt": f"full computeLighting_optimized: GI {args.spp} spp x 3 bounces + 4 soft shadow rays per vertex",
                "frag-web": "web/shader.wgsl computeLighting_optimized: GI 32 x 3 bounces, 8-tap Poisson-disk light per vertex",
                "frag-root": "shader.wgsl computeLighting_optimized: 16 fixed directions x 1 bounce (roulette 0.9), 8-tap Poisson-disk light"}[args.mode]
        scene_txt = {"s1m": "cfg3: S1M closed-form icosphere stand-in for dragon.obj",
                     "s70k": "cfg1-2: S70k closed-form icosphere stand-in for bunny.obj"}.get(
                         args.scene, f"mesh file {os.path.basename(args.scene[4:])}" if from_file else f"cfg5: {args.scene[7:]} copies of S1M on a lattice, flattened")
        out = {
            "metric": "Mrays/s primary+shadow, dragon.obj 4K" if (args.mode == "shadow" and args.scene == "s1m" and (W, H) == (3840, 2160))
                      else f"Mrays/s {args.mode} {args.scene} {W}x{H}",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_txt} ({build['n_tri']} triangles), {W}x{H}, {what}, camera B (0,0,1.3), "
                                   f"layout {args.layout}, node records {'v2' if encoding == rt.NODES_V2 else 'v1'}, "
                                   + ((f"textured shading: the file's {len(file_mesh.materials)} materials, map_Kd images bound: {[n for n, _ in bound] or 'none found (Kd only)'}, nearest texel"
                                       if from_file else "textured shading: 3 materials, map_Kd 2048^2 + 1024^2 closed-form RGB8, nearest texel") if textured
                                      else "constant albedo (compute.wgsl:372)"),
                       "mode": args.mode, "rays_per_frame": int(total_rays), "rays_traced_per_frame": int(total_traced), "frames_in_flight": fif, "frames_per_pipeline": batch, "stripes": "16-row round-robin" if world > 1 else "none",
                       "exchange": ("gather to rank 0" if args.exchange == "root" else "all_gather") if world > 1 else "none"},
            "lbvh_build_ms": {k: round(float(v), 4) for k, v in build.items() if k.startswith("ms_")},
            "lbvh_build_wall_ms": round(build_wall_ms, 4),     # host clock around rt_build_lbvh (launches + events + the read-back)
            "lbvh_build_roofline_frac": round(build["algorithmic_bytes"] / (build["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "bvh_broadcast_ms": round(bcast_ms, 3),
            "frame_kernels_ms": round(kernel_ms, 4), "frame_kernels_ms_max_rank": round(kernel_max, 4),
            # rays that actually enter the tree (k_render's input): `value` also counts one primary ray for every pixel the
            # cull pass disposes of, as the reference's per-pixel loop casts it (compute.wgsl:268-275)
            "value_traced": round(total_traced * args.steps / elapsed / 1e6, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc.get("k_render_hbm_bytes_per_launch"),
                         "traffic_source": pmc.get("source"),
                         "kernel": "k_render", "kernel_ms": round(trav_ms, 4), "algorithmic_bytes_per_launch": int(my_bytes * frames_per_launch),
                         "frames_per_launch": round(frames_per_launch, 3),
                         "bytes_outside_kernel": int(my_frame_bytes - my_bytes),
                         # what actually binds the kernel: its working set is cache-resident, fabric traffic is a
                         # small share of the HBM roof, and the VALU lanes are half idle behind divergent 64-byte gathers
                         "fabric_frac": (round(pmc["k_render_hbm_bytes_per_launch"] / (trav_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                         if pmc.get("k_render_hbm_bytes_per_launch") and trav_ms > 0 else None),
                         "valu_lane_util": pmc.get("k_render_valu_lane_util"),
                         "note": "achieved = SURVEY 8d's algorithmic bytes (32 B per box test, 48 per triangle test, 48 per hit update, 4 per pixel) / "
                                 "kernel time; the records come out of L2 / Infinity Cache (fabric traffic = `traffic`, ~17x less), so frac can "
                                 "pass 1: the kernel is bound by dependent 64-byte gathers, not by HBM"},
        }
        if frame_check is not None:
            out["assembled_frame_vs_single_gpu_render"] = frame_check
        if world > 1 and bframes is not None and bframes.windowed and bframes.window[0] is not None:
            (col0, n_cols, g0, n_rows), _ = bframes.window[0]
            out["config"]["exchange"] += (f"; only the frames' screen rectangle travels: columns {col0}..{col0 + n_cols - 1} of packed rows "
                                          f"{g0 * rdist.STRIPE_ROWS}..{g0 * rdist.STRIPE_ROWS + n_rows - 1} of every rank = "
                                          f"{n_rows * n_cols * 4 * world / 1e6:.1f} MB per frame instead of {frame.rpr * W * 4 * world / 1e6:.1f} MB")
        if args.rehearse_on_one_gpu and world > 1:
            out["rehearsal"] = f"{world} ranks sharing cuda:0, collectives over gloo: exercises the N > 1 code path, NOT a measurement"
        if legs:
            out["legs"] = legs
        if fif2 is not None:
            out["two_pipelines_in_flight" if batch > 1 else "two_frames_in_flight"] = fif2
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, mesh, U, mode, textured, shade)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def pmc_traffic(args, world, textured, batch, frames_per_launch):
    """HBM bytes per k_render launch from the PMC passes (scripts/pmc_passes.sh -> scripts/pmc_summary.py,
    2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md), committed as profiles/traffic.json.  The counters
    need rocprofv3 runs of their own, so the figure is read back from the last committed collection and is
    only reported when that collection was made on this exact workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return {}
    w = t.get("workload", {})
    same = (world == 1 and w.get("scene") == args.scene and w.get("width") == args.width and w.get("height") == args.height
            and w.get("mode") == args.mode and w.get("layout") == args.layout and (args.mode != "pt" or w.get("spp") == args.spp)
            and bool(w.get("textured", False)) == textured)
    if not same:
        return {}
    if int(w.get("frames_per_pipeline", 1)) != batch:
        return {}
    # counters are per launch of the collection run; this run's launches may hold another number of frames (K = 20 -> 10 + 10)
    per_frame = t.get("k_render_hbm_bytes_per_launch", 0) / max(float(t.get("frames_per_launch", 1.0)), 1e-9)
    return {"k_render_hbm_bytes_per_launch": int(per_frame * frames_per_launch), "k_render_valu_lane_util": t.get("k_render_valu_lane_util"),
            "source": "profiles/traffic.json: " + str(t.get("source", ""))}


def cpu_baseline(args, mesh, U, mode, textured, shade):
    """The scalar oracle ("port") on the host cores, on a bounded sample of the same frame:
    every k-th row of the 4K image, k chosen so that the sample takes ~10-30 s of CPU time."""
    from oracle import binding as oracle     # checker / baseline only — never on the product path
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
    enc = 2 if 2 * mesh.n_tri - 1 > (1 << 24) or args.encoding == "v2" else 1
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, encoding=enc)
    surf = None
    if textured:
        from raytracer_amd import scenes
        surf = oracle.Surface(oracle.gather_tri_uv(mesh.texcoords, mesh.tex_idx, mesh.material_ids, b["order"]), scenes.STANDIN_MATERIALS,
                              [scenes.procedural_texture(2048, 0), scenes.procedural_texture(1024, 1)])
    H = args.height
    # probe: 8 rows spread over the frame
    _, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, **shade, rows=(0, H, max(1, H // 8)),
                          n_threads=cores, want=(), encoding=enc, surface=surf)
    per_row = max(st["seconds"] / max(1, len(range(0, H, max(1, H // 8)))), 1e-6)
    n_rows = int(min(H, max(8, args.cpu_seconds / per_row)))
    step = max(1, H // n_rows)
    # repeat the sample until ~cpu_seconds of CPU work (wall x threads) has been spent, at least twice
    rays = 0
    secs = 0.0
    reps = 0
    while reps < 2 or (secs * cores < args.cpu_seconds and reps < 64):
        _, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, **shade, rows=(0, H, step),
                              n_threads=cores, want=(), encoding=enc, surface=surf)
        rays += st["rays_primary"] + st["rays_shadow"] + st["rays_bounce"]
        secs += st["seconds"]
        reps += 1
    # the same oracle on ONE thread, on every 16th row of that sample (~1/16 of its work)
    _, s1 = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, **shade, rows=(0, H, step * 16),
                          n_threads=1, want=(), encoding=enc, surface=surf)
    single = (s1["rays_primary"] + s1["rays_shadow"] + s1["rays_bounce"]) / max(s1["seconds"], 1e-9) / 1e6
    return {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "single_thread_value": round(single, 3),
            "sample": f"rows 0..{H} step {step} of the same {args.width}x{H} frame ({len(range(0, H, step))} rows) x {reps} "
                      f"repeats: {rays} rays in {secs:.2f} s wall on {cores} threads ({secs * cores:.1f} s of CPU work)",
            "lbvh_build_ms": round(b["seconds"] * 1e3, 2), "lbvh_build_cores": 1}


if __name__ == "__main__":
    main()
