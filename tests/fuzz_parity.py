#!/usr/bin/env python3
"""Randomised parity soak: seeded random meshes (soups with duplicate Morton codes, degenerate and huge triangles, tiny and
block-boundary sizes, shared-vertex grids), random cameras / modes (the two older fragment shaders included) / textures / stack depths / layouts / encodings / stripe shares /
batch sizes, the no-alloc entry point — LBVH build and frames of the HIP path against the CPU oracle, bit for bit.  usage: python tests/fuzz_parity.py [n_cases] [seed0]   (test infrastructure: tests/test_gpu_fuzz.py runs a short, fixed-seed soak)
Prints one line per case; exits 1 at the first mismatch (with the case's parameters, which reproduce it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
from oracle import binding as oracle      # the checker (this file is test infrastructure)

dev = torch.device("cuda:0")


def make_mesh(rng, kind, T):
    if kind == "grid":                                   # shared vertices, smooth normals, a displaced height field
        n = max(2, int(np.sqrt(T / 2)) + 1)
        u, v = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
        z = 0.3 * np.sin(3 * u + rng.random()) * np.cos(2 * v + rng.random())
        pos = np.stack([u, v, z], -1).reshape(-1, 3).astype(np.float32)
        nor = np.tile(np.float32([0, 0, 1]), (pos.shape[0], 1))
        i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
        a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
        idx = np.stack([a, b, c, a, c, d], -1).reshape(-1).astype(np.uint32)
        return rt.Mesh(pos, nor, idx, idx.copy())
    c = rng.random((T, 1, 3))
    if kind == "dups":
        q = int(rng.integers(1, 6))
        c = np.floor(c * q) / q                          # many equal centroids -> equal codes -> the stable order decides
    size = {"soup": 0.2, "dups": 0.1, "big": 1.5, "flat": 0.3}[kind]
    p = c + (rng.random((T, 3, 3)) - 0.5) * size
    if kind == "flat":
        p[:, :, int(rng.integers(0, 3))] = 0.25          # zero extent on one axis: invExtent = 0 there
    if T > 4:
        p[rng.integers(0, T)] = p[rng.integers(0, T)]    # an exact duplicate triangle
        p[rng.integers(0, T), 2] = p[rng.integers(0, T), 1]   # a degenerate one (two equal corners)
    pos = p.reshape(-1, 3).astype(np.float32)
    nor = rng.standard_normal((T * 3, 3)).astype(np.float32)
    nor /= np.linalg.norm(nor, axis=1, keepdims=True)
    idx = np.arange(T * 3, dtype=np.uint32)
    return rt.Mesh(pos, nor, idx, idx.copy())


def fail(msg, params):
    raise AssertionError(f"MISMATCH: {msg} {params}")


def run(n_cases, seed0, log=print):
  """-> mean hit fraction; raises AssertionError (with the parameters that reproduce it) at the first mismatch."""
  oracle.build()
  t_start = time.time()
  all_hits = []
  for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    kind = ["soup", "dups", "big", "flat", "grid"][case % 5]
    T = int(rng.choice([1, 2, 3, 7, 255, 256, 257, 511, 513, 1000, 4097, 20000, 70000]))
    mesh = rt.normalize_mesh(make_mesh(rng, kind, T))
    T = mesh.n_tri
    enc = int(rng.choice([rt.NODES_V1, rt.NODES_V2]))
    mode = int(rng.choice([rt.MODE_PRIMARY, rt.MODE_SHADOW, rt.MODE_PT, rt.MODE_SHADOW, rt.MODE_FRAG_WEB, rt.MODE_FRAG_ROOT]))
    frag = mode in (rt.MODE_FRAG_WEB, rt.MODE_FRAG_ROOT)
    textured = (not frag) and bool(rng.integers(0, 2))
    layout = rt.LAYOUT_WIDE64 if frag else int(rng.choice([rt.LAYOUT_WIDE64, rt.LAYOUT_AOS32]))
    stack = int(rng.choice([2, 3, 5, 12, 32]))
    w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
    gi, nb, soft = int(rng.integers(1, 5)), int(rng.integers(0, 4)), int(rng.integers(1, 6))
    if frag:
        w, h, gi, nb = min(w, 48), min(h, 40), int(rng.integers(1, 4)), min(nb, 2)      # the fragment shaders' GI loops are long
    n_frames = int(rng.integers(1, 33)) if case % 4 == 0 else int(rng.integers(1, 17))
    world = int(rng.choice([1, 1, 2, 3]))
    rank = int(rng.integers(0, world))
    params = dict(case=case, seed=seed0 + case, kind=kind, T=T, enc=enc, mode=mode, layout=layout, stack=stack, w=w, h=h, gi=gi, nb=nb,
                  soft=soft, n_frames=n_frames, world=world, rank=rank, textured=textured)
    ctx = rt.Context(0)
    ctx.set_node_encoding(enc)
    ctx.set_mesh(mesh)
    surf = None
    if textured:      # random texcoords per corner, material ids incl. "none" (-1) and out of range, a material without a texture
        mesh.texcoords = rng.uniform(-2.0, 3.0, (max(3, T), 2)).astype(np.float32)
        mesh.tex_idx = rng.integers(0, mesh.texcoords.shape[0], 3 * T).astype(np.uint32)
        mesh.material_ids = rng.integers(-1, 4, T).astype(np.int32)
        mats = [((1.0, 1.0, 1.0), 0), ((0.9, 0.5, 0.2), -1), ((0.6, 0.9, 1.0), 1)]
        texs = [rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 3)).astype(np.uint8) for _ in range(2)]
        ctx.set_surface(mesh)
        ctx.set_materials(mats)
        for i, t in enumerate(texs):
            ctx.set_texture(i, t)
    st = ctx.build_lbvh()
    ob = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, encoding=enc)
    f = ctx.flatten_for_ssbo()
    if not np.array_equal(ctx.morton(), ob["codes"]): fail("codes", params)
    if not np.array_equal(ctx.tri_order(), ob["order"]): fail("order", params)
    if not np.array_equal(f["bvhData"].view(np.uint32), ob["nodes"].view(np.uint32)): fail("nodes", params)
    if not np.array_equal(f["triPos"].view(np.uint32), ob["tri_pos"].view(np.uint32)): fail("tri_pos", params)
    if not np.array_equal(f["triNor"].view(np.uint32), ob["tri_nor"].view(np.uint32)): fail("tri_nor", params)
    if st["refit_fallbacks"] != 0: fail("refit_fallbacks", params)
    if textured:
        tri_uv = oracle.gather_tri_uv(mesh.texcoords, mesh.tex_idx, mesh.material_ids, ob["order"])
        if not np.array_equal(ctx.tri_uv().view(np.uint32), tri_uv.view(np.uint32)): fail("tri_uv", params)
        surf = oracle.Surface(tri_uv, mats, texs)
    Us = [rt.make_uniforms(w, h, cam_pos=tuple(rng.normal(0, 0.3, 3) + np.float64([0, 0, 1.4])), yaw=float(rng.normal(0, 0.2)),
                           pitch=float(rng.normal(0, 0.2)), rot_deg=tuple(rng.uniform(-180, 180, 3)), scale=(2.2, 2.2, 2.2),
                           time=float(rng.integers(0, 50)))
          for _ in range(n_frames)]
    so = dict(stripe_rows=8 * int(rng.integers(1, 3)), stripe_rank=rank, stripe_world=world) if world > 1 else {}
    opts = rt.render_opts(mode=mode, gi_samples=gi, gi_bounces=nb, soft_samples=soft, max_stack=stack, layout=layout,
                          flags=rt.FLAG_TEXTURED if textured else 0, **so)
    rows = rt.packed_rows(h, so["stripe_rows"], rank, world) if world > 1 else h
    ys = (np.array([y for y in range(h) if (y // so["stripe_rows"]) % world == rank], dtype=np.int64) if world > 1 else np.arange(h))
    refs = []
    for U in Us:
        r, _ = oracle.render(ob["nodes"], ob["tri_pos"], ob["tri_nor"], U, mode=mode, gi_samples=gi, gi_bounces=nb, soft_samples=soft,
                             max_stack=stack, n_threads=8, encoding=enc, surface=surf)
        refs.append(r)
    bufs = [torch.zeros((n_frames, max(rows, 1), w), dtype=torch.int32, device=dev) for _ in range(3)]
    bufs[2] = bufs[2].view(torch.float32)
    if rows > 0:
        for rep in range(2):                              # second pass: with the hints the first one left
            for f_ in range(n_frames):                    # per frame
                ctx.render_device(Us[f_], opts, bufs[0][f_].data_ptr(), bufs[1][f_].data_ptr(), bufs[2][f_].data_ptr(), want_stats=False)
            ctx.synchronize()
            got = [b.cpu().numpy() for b in bufs]
            for f_ in range(n_frames):
                if not np.array_equal(got[1][f_][:rows].view(np.uint32), refs[f_]["hit_tri"][ys]): fail(f"per-frame hit_tri f={f_} rep={rep}", params)
                if not np.array_equal(got[2][f_][:rows].view(np.uint32), refs[f_]["hit_t"][ys].view(np.uint32)): fail(f"per-frame hit_t f={f_}", params)
                if not np.array_equal(got[0][f_][:rows].view(np.uint8).reshape(rows, w, 4), refs[f_]["rgba8"][ys]): fail(f"per-frame rgba8 f={f_}", params)
            for b in bufs: b.zero_()
            for f_ in range(min(n_frames, 2)):            # the entry point that allocates nothing: same frames
                ctx.render_device_noalloc(Us[f_], opts, bufs[0][f_].data_ptr(), bufs[1][f_].data_ptr(), bufs[2][f_].data_ptr())
            ctx.synchronize()
            got = [b.cpu().numpy() for b in bufs]
            for f_ in range(min(n_frames, 2)):
                if not np.array_equal(got[1][f_][:rows].view(np.uint32), refs[f_]["hit_tri"][ys]): fail(f"noalloc hit_tri f={f_} rep={rep}", params)
                if not np.array_equal(got[2][f_][:rows].view(np.uint32), refs[f_]["hit_t"][ys].view(np.uint32)): fail(f"noalloc hit_t f={f_}", params)
                if not np.array_equal(got[0][f_][:rows].view(np.uint8).reshape(rows, w, 4), refs[f_]["rgba8"][ys]): fail(f"noalloc rgba8 f={f_}", params)
            if frag:
                continue                                  # batches: the compute.wgsl kind only
            for b in bufs: b.zero_()
            ctx.render_batch_device(np.stack(Us), opts, [bufs[0][f_].data_ptr() for f_ in range(n_frames)],
                                    [bufs[1][f_].data_ptr() for f_ in range(n_frames)], [bufs[2][f_].data_ptr() for f_ in range(n_frames)])
            ctx.synchronize()
            got = [b.cpu().numpy() for b in bufs]
            for f_ in range(n_frames):
                if not np.array_equal(got[1][f_][:rows].view(np.uint32), refs[f_]["hit_tri"][ys]): fail(f"batch hit_tri f={f_} rep={rep}", params)
                if not np.array_equal(got[2][f_][:rows].view(np.uint32), refs[f_]["hit_t"][ys].view(np.uint32)): fail(f"batch hit_t f={f_}", params)
                if not np.array_equal(got[0][f_][:rows].view(np.uint8).reshape(rows, w, 4), refs[f_]["rgba8"][ys]): fail(f"batch rgba8 f={f_}", params)
    hits = float(np.mean([(r["hit_tri"] != 0xFFFFFFFF).mean() for r in refs]))
    all_hits.append(hits)
    log(f"case {case:3d} ok  {kind:5s} T={T:6d} enc={enc} mode={mode} tex={int(textured)} layout={layout} stack={stack:2d} {w}x{h} frames={n_frames} "
        f"world={world} hit {hits:.2f}  [{time.time() - t_start:.0f} s]")
    ctx.close()
  return float(np.mean(all_hits))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    try:
        mean_hit = run(n, s0, log=lambda m: print(m, flush=True))
    except AssertionError as e:
        print(e, flush=True)
        sys.exit(1)
    print("all", n, "cases identical; mean hit fraction", round(mean_hit, 3))
