#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: Mrays/s, primary + shadow rays, dragon-class
mesh at 3840x2160 (cfg 3), plus LBVH build ms.  `web/dragon.obj` is a Git-LFS pointer in the
reference, so the workload is the closed-form stand-in S1M (1 310 720 triangles), textured with the
reference's own base-colour images (models/Material_85_baseColor.jpg / Material_86_baseColor.png, tests/golden fixtures).

  python bench.py --gpus N --steps K --warmup W
  N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one frame(): the traversal kernel over the whole 4K image (inputs resident in HBM).  The K timed frames are K
DIFFERENT views - the model turns by one degree per frame, as a user dragging the reference's rotY slider - queued as
pipelines of up to 32 frames (rt_render_batch_device).  For N > 1 every frame is sharded as 16-row image-space stripes +
RCCL gather of the packed stripes + de-interleave, all inside the timed region; the BVH is built once on rank 0 and
replicated with RCCL before the timed region.  Total work is fixed as N grows => "strong".

One JSON line on rank 0.  `value` counts the rays that ENTER the traversal kernel ("rays actually cast / kernel
time"): primary rays that miss the tree's root box are disposed of by the cull pass and are only in `value_all_pixels`.
Rays and work counters are the kernels' own, read from a counted run of exactly the timed pipelines (rt_batch_stats;
tests/test_gpu_batch.py proves them equal to the oracle's per frame).  `roofline` prices the traversal kernel against the
memory level that serves its bytes (PMC: L2 hits / L2 misses = Infinity Cache + HBM / fabric bytes) and names the binding
one; the SURVEY 8d algorithmic bytes (32 V + 48 K + 48 H + 4 P) are reported beside it.  `cpu_baseline` is the scalar
oracle timed on the host cores on a bounded sample of the same frames (reported, not a target).
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
    ap.add_argument("--no-fif2", action="store_true", help="(accepted for old scripts; the two-pipelines-in-flight loop is gone)")
    ap.add_argument("--no-legs", action="store_true", help="skip the per-frame and static-view legs at N = 1")
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
        # the stage breakdown comes from builds of its own: the events between the launches cost the build ~2 us each
        ctx.set_build_stage_timing(True)
        staged = sorted((ctx.build_lbvh() for _ in range(3)), key=lambda s: s["ms_total"])[1]
        ctx.set_build_stage_timing(False)
        build = dict(build, **{k: v for k, v in staged.items() if k.startswith("ms_") and k not in ("ms_total", "ms_pack")}, ms_total_with_stage_events=staged["ms_total"])
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
    elif textured:                      # materials + the reference's base-colour images: small host-side data, set on every rank
        from raytracer_amd import scenes
        ctx.set_materials(scenes.STANDIN_MATERIALS)
        for slot, img in enumerate(cfg3_textures()):
            ctx.set_texture(slot, img)

    # camera (B) of SURVEY 8d: pos (0,0,1.3), default sliders; frame k of a loop shows the model turned by k degrees (rotY)
    def view(k):
        return rt.make_uniforms(W, H, cam_pos=(0.0, 0.0, 1.3), rot_deg=(0.0, 1.0 * k, 0.0))
    U = view(0)
    K = args.steps
    views = [view(k) for k in range(K)]
    shade = rt.render_opts_for_mode(mode)      # sample counts / stack depth the shader file is written with
    if not frag:
        shade.gi_samples = args.spp
    shade = dict(gi_samples=shade.gi_samples, gi_bounces=shade.gi_bounces, soft_samples=shade.soft_samples, max_stack=shade.max_stack)
    base = rt.render_opts(mode=mode, layout=layout, flags=tflag, **shade)
    # auto: 32 (RT_MAX_BATCH) on one GPU - K = 20 steps are ONE pipeline; 16 with N > 1, where two pipelines (10 + 10) let the
    # exchange of the first overlap the traversal of the second
    batch = (32 if world == 1 else 16) if args.batch == "auto" else int(args.batch)
    # pipelines (frames, or batches of frames) in flight: 1 at N = 1, so that every k_render launch runs alone and its duration is
    # the kernel's own.  Batches stay at 1 with N > 1 too: a second batch in flight gains 5 % on a 1/8 share of a 4K frame and
    # loses 27 % on an 8K frame (each pipeline then owns 2 of the 3 resident blocks per CU); single frames use three
    fif = (1 if (world == 1 or batch > 1) else 3) if args.frames_in_flight == "auto" else int(args.frames_in_flight)
    frame = rdist.ShardedFrame(rdist.GpuBackend(ctx, device, frames_in_flight=fif), W, H, rank, world, base, exchange=args.exchange)
    bframes = rdist.BatchedFrames(frame.b, W, H, rank, world, batch, base, exchange=args.exchange) if batch > 1 else None

    def run_frames(vs, n, per_frame=False):
        """n steps (frames) of the frame loop, vs[i % len(vs)] each: one frame per pipeline, or `batch` frames per pipeline."""
        if bframes is None or per_frame:
            for i in range(n):
                frame.step(vs[i % len(vs)], want_stats=False)
            return
        i0 = 0
        for size in rdist.batch_sizes(n, batch):             # pipelines of equal size (K = 20, 16 at most per pipeline: 10 + 10)
            bframes.step(np.stack([vs[i % len(vs)] for i in range(i0, i0 + size)]))
            i0 += size

    # Rays and work counters of the n frames of a loop (this rank's share), from the kernels' OWN counters of exactly the
    # pipelines the loop runs: the same batches once more with RT_FLAG_COUNT_WORK (rt_batch_stats), untimed.
    copts_full = rt.render_opts(mode=mode, layout=layout, flags=rt.FLAG_COUNT_WORK | tflag, **shade,
                                stripe_rows=frame.opts.stripe_rows, stripe_rank=rank, stripe_world=world)
    scratch_img = torch.zeros((batch, frame.rpr if world > 1 else H, W), dtype=torch.int32, device=device)

    popts = rt.render_opts(mode=rt.MODE_PRIMARY, layout=layout, flags=rt.FLAG_COUNT_WORK | tflag, **shade,
                           stripe_rows=frame.opts.stripe_rows, stripe_rank=rank, stripe_world=world)

    def counted(vs, n, per_frame=False, primary_only=False):
        """-> per frame (rays as the reference's loop casts them, rays that enter the tree, SURVEY 8d bytes of the whole frame,
        k_render's own bytes), and the counting kernels' average launch time.  Pixels whose primary ray misses the root box never
        reach k_render (k_cull disposes of them): each is one ray + one box test + one pixel, and those 36 B are NOT k_render's."""
        per = []
        copts = popts if primary_only else copts_full
        ctx.profile_begin()
        if bframes is None or per_frame:
            sts = [ctx.render_device(vs[i % len(vs)], copts, d_rgba8=scratch_img[0].data_ptr()) for i in range(n)]
        else:
            sts, i0 = [], 0
            for size in rdist.batch_sizes(n, batch):
                ctx.render_batch_device(np.stack([vs[i % len(vs)] for i in range(i0, i0 + size)]), copts,
                                        [scratch_img[f].data_ptr() for f in range(size)])
                sts += ctx.batch_stats(size)
                i0 += size
        _, cms, _ = ctx.profile_end()
        for c in sts:
            culled = c["pixels"] - c["pixels_traced"]
            rays = c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
            fb = 32 * c["box_tests"] + 48 * c["tri_tests"] + 48 * c["hit_updates"] + 4 * c["pixels"]
            per.append((rays, rays - culled, fb, fb - 36 * culled))
        return per, cms

    if bframes is not None:             # set-up (untimed): one pipeline of the largest size the loop uses sizes the scratch
        bframes.step(np.stack([view(k) for k in range(max(rdist.batch_sizes(K, batch)))]))
        bframes.finish()
    # warm-up: W steps - in whole pipelines of the timed loop's shape when frames are batched, so that every launch of the traversal
    # kernel in a run has the timed launches' size (rocprofv3's per-kernel average is then the average of like launches)
    # (profiled, so that the ring of hipEvents rt_profile_begin creates on first use exists before the clock starts: 1280
    # hipEventCreate calls cost the first timed pipeline 0.15 ms)
    ctx.profile_begin()
    run_frames(views, args.warmup if bframes is None else -(-args.warmup // K) * K)
    ctx.profile_end_phases()

    def fence():
        frame.finish()
        if bframes is not None:
            bframes.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(vs, n, per_frame=False):
        """The loop under the clock: barrier + synchronize on both sides; hipEvents recorded on the context's own stream around
        every pipeline's kernels (rt_profile_begin/end) give the kernel durations of exactly these n steps."""
        fence()
        t0 = time.perf_counter()
        ctx.profile_begin()
        run_frames(vs, n, per_frame)
        t_enq = time.perf_counter()
        fence()
        e = time.perf_counter() - t0
        if os.environ.get("RT_BENCH_TRACE"):
            print(f"[bench trace] enqueue {1e3 * (t_enq - t0):.3f} ms, wait {1e3 * (time.perf_counter() - t_enq):.3f} ms, total {1e3 * e:.3f} ms", file=sys.stderr, flush=True)
        # per pipeline: all its kernels (cull, scan, compact, traversal) / the traversal launch / the shade launch (0 unless split)
        pipe_ms, trav_ms, shade_ms, launches = ctx.profile_end_phases()
        return e, pipe_ms, trav_ms, shade_ms, launches

    elapsed, kernel_ms, trav_ms, shade_ms, n_launch = timed(views, K)
    frames_per_launch = K / max(n_launch, 1)               # 1, or the mean batch size (a k_render launch then walks that many frames' lists)
    per, counted_ms = counted(views, K)
    my_rays, my_rays_traced = sum(p[0] for p in per), sum(p[1] for p in per)
    my_frame_bytes, my_bytes = sum(p[2] for p in per), sum(p[3] for p in per)
    # RT_MODE_SHADOW pipelines of a batch run as two launches: the walk of the primary rays (exactly RT_MODE_PRIMARY's box / triangle /
    # hit counters) and the shade launch (the shadow rays' box tests): the first one's own algorithmic bytes from a primary-only count
    split = bframes is not None and mode == rt.MODE_SHADOW and shade_ms > 0.0
    trav_bytes = sum(p[3] for p in counted(views, K, primary_only=True)[0]) if split else my_bytes

    # N > 1: the frame rank 0 assembled from all ranks' stripes (the last timed one) against rank 0's own render of the whole
    # frame — outside the timed region; says on the driver's 8-GPU run that the exchange delivered the right pixels
    frame_check = None
    if world > 1 and rank == 0:
        try:
            if bframes is not None:     # (the counted run wrote into its own buffer: bframes.images still holds the timed loop's last pipeline)
                first_of_last = K - rdist.batch_sizes(K, batch)[-1]
                assembled, shown = bframes.images[0], views[first_of_last]
            else:
                assembled, shown = frame.image, views[K - 1]
            whole = torch.zeros((H, W), dtype=torch.int32, device=device)
            ctx.render_device(shown, base, d_rgba8=whole.data_ptr(), want_stats=False)
            ctx.synchronize()
            frame_check = "identical" if torch.equal(whole, assembled) else f"{int((whole != assembled).sum())} pixels differ"
        except Exception as e:          # (a check outside the timed region must not cost the line)
            frame_check = f"check failed: {type(e).__name__}: {e}"

    # Legs at N = 1, reported beside `value` (never as `value`): the same camera path with ONE frame per pipeline (the
    # latency-bound loop of an interactive viewer, what the reference's frame() is), and a STATIC view in batches (K copies of
    # view 0: what round 2 reported as `value` - the copies share cache lines and run with converged scheduling hints).
    legs = {}
    if world == 1 and not args.no_legs:
        def leg(vs, per_frame):
            run_frames(vs, max(args.warmup, batch if not per_frame else 3), per_frame)
            e, _, tr_ms, sh_ms, launches = timed(vs, K, per_frame)
            pr, _ = counted(vs, K, per_frame)
            traced, byts = sum(p[1] for p in pr), sum(p[3] for p in pr)
            return {"value": round(traced / e / 1e6, 2), "unit": "Mrays/s", "value_all_pixels": round(sum(p[0] for p in pr) / e / 1e6, 2),
                    "ms_per_step": round(e / K * 1e3, 4), "kernel_ms": round(tr_ms + sh_ms, 4), "frames_per_launch": round(K / max(launches, 1), 3),
                    "algorithmic_gbs": round(byts / ((tr_ms + sh_ms) * launches * 1e-3) / 1e9, 1) if tr_ms > 0 else None}
        if batch > 1:
            legs["per_frame"] = leg(views, True)
            legs["per_frame"]["what"] = ("the same camera path, ONE frame per pipeline (rt_render_device, one frame in flight): the latency-bound loop of "
                                         "an interactive viewer; scheduling hints with tile-level promotion under motion")
            legs["static_view"] = leg([U], False)
            legs["static_view"]["what"] = (f"{K} copies of view 0 in the same pipelines (round 2's headline loop): identical frames share cache "
                                           "lines and are ordered by converged scheduling hints - not a workload, kept for comparison")
        run_frames(views, batch)        # leave the context as the main loop had it
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
        ms_per_step = elapsed / K * 1e3
        value = total_traced / elapsed / 1e6
        launch_bytes = trav_bytes / max(n_launch, 1)       # this rank's algorithmic bytes per launch of the traversal kernel
        what = {"primary": "primary rays only", "shadow": "primary + 4 soft shadow rays",
                "pt": f"full computeLighting_optimized: GI {args.spp} spp x 3 bounces + 4 soft shadow rays per vertex",
                "frag-web": "web/shader.wgsl computeLighting_optimized: GI 32 x 3 bounces, 8-tap Poisson-disk light per vertex",
                "frag-root": "shader.wgsl computeLighting_optimized: 16 fixed directions x 1 bounce (roulette 0.9), 8-tap Poisson-disk light"}[args.mode]
        scene_txt = {"s1m": "cfg3: S1M closed-form icosphere stand-in for dragon.obj",
                     "s70k": "cfg1-2: S70k closed-form icosphere stand-in for bunny.obj"}.get(
                         args.scene, f"mesh file {os.path.basename(args.scene[4:])}" if from_file else f"cfg5: {args.scene[7:]} copies of S1M on a lattice, flattened")
        tex_txt = ((f"textured shading: the file's {len(file_mesh.materials)} materials, map_Kd images bound: {[n for n, _ in bound] or 'none found (Kd only)'}, nearest texel"
                    if from_file else "textured shading: 3 materials, map_Kd = the reference's models/Material_85_baseColor.jpg and Material_86_baseColor.png "
                                      "(tests/golden fixtures: box-filtered to 512^2, replicated back to their 2048^2 texels), nearest texel") if textured
                   else "constant albedo (compute.wgsl:372)")
        out = {
            "metric": "Mrays/s primary+shadow, dragon.obj 4K" if (args.mode == "shadow" and args.scene == "s1m" and (W, H) == (3840, 2160))
                      else f"Mrays/s {args.mode} {args.scene} {W}x{H}",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_txt} ({build['n_tri']} triangles), {W}x{H}, {what}, camera B (0,0,1.3), a camera path of {K} DIFFERENT views "
                                   f"(the model turns 1 degree per frame), {batch} frame(s) per pipeline at most, layout {args.layout}, "
                                   f"node records {'v2' if encoding == rt.NODES_V2 else 'v1'}, {tex_txt}",
                       "mode": args.mode, "rays_traced_per_frame": int(total_traced / K), "rays_all_pixels_per_frame": int(total_rays / K),
                       "frames_in_flight": fif, "frames_per_pipeline": batch, "stripes": "16-row round-robin" if world > 1 else "none",
                       "exchange": ("gather to rank 0" if args.exchange == "root" else "all_gather") if world > 1 else "none"},
            # `value` = rays that ENTER the traversal kernel (k_render's input); value_all_pixels also counts one primary ray for every
            # pixel the cull pass disposes of, as the reference's per-pixel loop casts it (compute.wgsl:268-275)
            "value_all_pixels": round(total_rays / elapsed / 1e6, 2),
            "lbvh_build_ms": {k: round(float(v), 4) for k, v in build.items() if k.startswith("ms_")},
            "lbvh_build_wall_ms": round(build_wall_ms, 4),     # host clock around rt_build_lbvh (launches + events + the read-back)
            "lbvh_build_roofline_frac": round(build["algorithmic_bytes"] / (build["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "bvh_broadcast_ms": round(bcast_ms, 3),
            "frame_kernels_ms": round(kernel_ms, 4), "frame_kernels_ms_max_rank": round(kernel_max, 4),
            "roofline": roofline(args, world, textured, batch, frames_per_launch, trav_ms, launch_bytes, counted_ms,
                                 int(my_frame_bytes - my_bytes) // max(n_launch, 1), shade_ms, int(my_bytes - trav_bytes) // max(n_launch, 1)),
        }
        if "per_frame" in legs:
            out["value_per_frame"] = legs["per_frame"]["value"]      # one frame per pipeline: comparable with round 1's loop
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
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, mesh, views, mode, textured, shade)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def cfg3_textures():
    """The reference's two base-colour images of models/plane_material.mtl (the only map_Kd images its tree holds next to a mesh
    that is present; SURVEY 8d row 3 names Material_85_baseColor.jpg), from the fixtures tests/golden/make_textures.py wrote:
    box-filtered 2048^2 -> 512^2 there, replicated here 4 x 4 back to 2048^2 texels so that the texture footprint in memory is
    the original's.  -> [texture slot 0, texture slot 1] as uint8 [2048, 2048, 3]."""
    import gzip
    out = []
    for name in ("Material_85_baseColor_512.ppm.gz", "Material_86_baseColor_512.ppm.gz"):
        raw = gzip.open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        assert raw.startswith(b"P6\n512 512\n255\n")
        img = np.frombuffer(raw, np.uint8, 512 * 512 * 3, len(b"P6\n512 512\n255\n")).reshape(512, 512, 3)
        out.append(np.ascontiguousarray(np.repeat(np.repeat(img, 4, axis=0), 4, axis=1)))
    return out


# peaks of the memory levels (MI355X_MICROARCH.md): L2 ~34.5 TB/s aggregate; random rows of an Infinity-Cache-resident table of
# k_render's size (wide 84 MB + tri_pre 63 MB: between the guide's 38 MB and 151 MB rows, 8.6 and 7.4-7.9 TB/s) - the higher
# figure is used, so the fraction errs low; HBM 8 TB/s
L2_PEAK_GBS, IC_PEAK_GBS = 34500.0, 8600.0


def roofline(args, world, textured, batch, frames_per_launch, trav_ms, launch_bytes, counted_ms, bytes_outside, shade_ms=0.0, shade_bytes=0):
    """The traversal kernel against the memory level that serves its bytes.  Time: the kernel's average launch duration over
    exactly the timed steps (hipEvents on its own stream).  Bytes per level: PMC counters of the same command, collected by
    rocprofv3 runs of their own (scripts/pmc_passes.sh -> scripts/pmc_summary.py -> profiles/traffic.json, committed), per frame
    of the collection's launches, scaled to this run's frames per launch:
        L2 hit bytes   = TCC_HIT_sum  x 128 B  (L2 line; calibrated by tools/pmc_calib on a known streaming read)  vs 34.5 TB/s
        L2 miss bytes  = TCC_MISS_sum x 128 B  (served by Infinity Cache or HBM)                                   vs  8.6 TB/s
        fabric bytes   = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the guide's gfx950 correction; Infinity-Cache hits are
                         counted in it, so an upper bound on HBM bytes)                                            vs  8.0 TB/s
    `bound` = the level with the largest fraction, `achieved` / `peak` / `frac` are that level's.  The SURVEY 8d algorithmic
    bytes (32 B per box test, 48 per triangle test, 48 per hit update, 4 per pixel; the kernel's own counters) are `algorithmic_*`:
    they are served out of L1 / L2 several times over, which is why they are NOT a byte roof of any level.
    Without a collection for this exact workload the levels are unknown: the algorithmic bytes are then priced against the L2
    peak (every one of them crosses L1 <- L2 at most once) and `traffic` is null."""
    t = trav_ms * 1e-3
    r = {"kernel": "k_render<KIND_TRAVERSE> (the primary rays' walk; the shade launch behind it is `shade_*`)" if shade_ms > 0 else "k_render",
         "kernel_ms": round(trav_ms, 4), "frames_per_launch": round(frames_per_launch, 3),
         "algorithmic_bytes_per_launch": int(launch_bytes), "algorithmic_gbs": round(launch_bytes / t / 1e9, 1) if t > 0 else None,
         "bytes_outside_kernel_per_launch": int(bytes_outside), "kernel_ms_counting": round(counted_ms, 4)}
    if shade_ms > 0:
        r["shade_kernel_ms"], r["shade_algorithmic_bytes_per_launch"] = round(shade_ms, 4), int(shade_bytes)
    pmc = pmc_levels(args, world, textured, batch)
    if not pmc or t <= 0:
        ach = launch_bytes / t / 1e9 if t > 0 else 0.0
        r.update({"bound": "l2", "achieved": round(ach, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(ach / L2_PEAK_GBS, 5), "traffic": None,
                  "note": "no PMC collection for this workload (profiles/traffic.json): algorithmic bytes against the L2 peak"})
        return r
    k = frames_per_launch                                  # counters are stored per FRAME of the collection's launches
    levels = {}
    for name, per_frame, peak in (("l2_hit", pmc["l2_hit_bytes_per_frame"], L2_PEAK_GBS), ("l2_miss", pmc["l2_miss_bytes_per_frame"], IC_PEAK_GBS),
                                  ("fabric", pmc["fabric_bytes_per_frame"], HBM_PEAK_GBS)):
        gbs = per_frame * k / t / 1e9
        levels[name] = {"bytes_per_launch": int(per_frame * k), "achieved": round(gbs, 1), "peak": peak, "frac": round(gbs / peak, 5)}
    bound = max(levels, key=lambda n: levels[n]["frac"])
    r.update({"bound": {"l2_hit": "l2", "l2_miss": "infinity_cache", "fabric": "hbm"}[bound], "achieved": levels[bound]["achieved"],
              "peak": levels[bound]["peak"], "unit": "GB/s", "frac": levels[bound]["frac"],
              "traffic": levels["fabric"]["bytes_per_launch"], "levels": levels,
              "l2_hit_rate": pmc.get("l2_hit_rate"), "valu_lane_util": pmc.get("valu_lane_util"), "ta_busy": pmc.get("ta_busy"),
              "valu_busy": pmc.get("valu_busy"), "wave_wait_share": pmc.get("wave_wait_share"), "shade": pmc.get("shade"),
              "traffic_source": pmc.get("source"),
              "busiest_unit": max((("valu", pmc.get("valu_busy") or 0.0), ("vector_memory_address_unit", pmc.get("ta_busy") or 0.0)), key=lambda u: u[1])[0],
              "note": "frac = the busiest memory level's bytes / kernel time / that level's peak (levels: L2 hits vs 34.5 TB/s, L2 misses vs the "
                      "8.6 TB/s random-row rate of the Infinity Cache, fabric bytes vs 8 TB/s HBM).  All three are far from their peaks: no memory level "
                      "bounds this kernel.  Its busiest unit is the VALU (`valu_busy` = 4 x SQ_INSTS_VALU / SIMD cycles, with `valu_lane_util` of the "
                      "lanes doing work: box tests diverge inside a wave), then the vector-memory address unit (`ta_busy`); the waves spend "
                      "`wave_wait_share` of their cycles waiting (chains of dependent record fetches).  profiles/r3/README.md has the formulas"})
    return r


def pmc_levels(args, world, textured, batch):
    """Per-frame PMC byte counts of k_render from the committed collection, only when it was made on this exact workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return {}
    w = t.get("workload", {})
    same = (world == 1 and w.get("scene") == args.scene and w.get("width") == args.width and w.get("height") == args.height
            and w.get("mode") == args.mode and w.get("layout") == args.layout and (args.mode != "pt" or w.get("spp") == args.spp)
            and bool(w.get("textured", False)) == textured and int(w.get("frames_per_pipeline", 1)) == batch
            and w.get("camera_path") == "rotY 1 degree per frame")
    if not same or "k_render" not in t:
        return {}
    return dict(t["k_render"], source="profiles/traffic.json: " + str(t.get("source", "")))


def cpu_baseline(args, mesh, views, mode, textured, shade):
    """The scalar oracle ("port") on the host cores, on a bounded sample of the same frames: every k-th row of the image,
    k chosen so that the sample takes ~10-30 s of CPU time, the repeats walking through the camera path's views."""
    U = views[0]
    from oracle import binding as oracle     # checker / baseline only — never on the product path
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
    enc = 2 if 2 * mesh.n_tri - 1 > (1 << 24) or args.encoding == "v2" else 1
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, encoding=enc)
    surf = None
    if textured:
        from raytracer_amd import scenes
        surf = oracle.Surface(oracle.gather_tri_uv(mesh.texcoords, mesh.tex_idx, mesh.material_ids, b["order"]), scenes.STANDIN_MATERIALS,
                              cfg3_textures())
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
        _, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], views[reps % len(views)], mode=mode, **shade, rows=(0, H, step),
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
            "what": "all rays the reference's loop casts (one per pixel + shadow / bounce rays): compare with value_all_pixels",
            "sample": f"rows 0..{H} step {step} of the first {min(reps, len(views))} views of the same {args.width}x{H} camera path "
                      f"({len(range(0, H, step))} rows each): {rays} rays in {secs:.2f} s wall on {cores} threads ({secs * cores:.1f} s of CPU work)",
            "lbvh_build_ms": round(b["seconds"] * 1e3, 2), "lbvh_build_cores": 1}


if __name__ == "__main__":
    main()
