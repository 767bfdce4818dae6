"""GPU: material / texture shading (RT_FLAG_TEXTURED, rtcore.h) against the oracle, through the C ABI.  The reference
has no textured shading: the oracle's definition is pinned by tests/test_oracle_texture.py's hand-derived
answers; here the HIP path must reproduce the oracle bit for bit.  Textures are closed-form stand-ins for the reference's
JPG/PNG files (models/), which cannot travel to the GPU box."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
COUNT_KEYS = ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests", "hit_updates", "pixels", "stack_drops")


@pytest.fixture(scope="module")
def plane_textured(rt, plane_obj_path):
    """web/plane.obj with a stand-in MTL: vt + usemtl through the tiny_obj_loader-compatible parser."""
    import shutil
    d = os.path.dirname(plane_obj_path)
    shutil.copy(os.path.join(GOLDEN, "plane_standin.mtl"), os.path.join(d, "plane_material.mtl"))   # the name plane.obj's mtllib asks for
    m = rt.normalize_mesh(rt.load_obj(plane_obj_path))
    assert len(m.materials) == 2 and set(np.unique(m.material_ids)) == {0, 1}
    assert m.materials[0]["map_Kd"] == "standin_checker_0.rgb" and m.texcoords.shape[0] >= 6088
    return m


def _bind(ctx, oracle, mesh, mats, texs):
    ctx.set_node_encoding(1)
    ctx.set_mesh(mesh)
    ctx.set_surface(mesh)
    ctx.build_lbvh()
    ctx.set_materials(mats)
    for i, t in enumerate(texs):
        ctx.set_texture(i, t)
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    tri_uv = oracle.gather_tri_uv(mesh.texcoords, mesh.tex_idx, mesh.material_ids, b["order"])
    assert np.array_equal(ctx.tri_uv().view(np.uint32), tri_uv.view(np.uint32))       # BVH-order records, bit for bit
    return b, oracle.Surface(tri_uv, mats, texs)


@pytest.mark.parametrize("layout", ["wide64", "aos32"])
@pytest.mark.parametrize("mode", ["primary", "shadow", "pt"])
def test_plane_textured(gpu_ctx, oracle, rt, plane_textured, mode, layout):
    from raytracer_amd import scenes
    m = plane_textured
    mats = [(tuple(x["diffuse"]), i) for i, x in enumerate(m.materials)]
    texs = [scenes.procedural_texture(256, 0), scenes.procedural_texture(64, 1)]
    b, surf = _bind(gpu_ctx, oracle, m, mats, texs)
    w, h = (160, 96) if mode != "pt" else (96, 64)
    U = rt.make_uniforms(w, h, **PLANE_VIEW)
    mid = dict(primary=0, shadow=1, pt=2)[mode]
    lay = dict(wide64=rt.LAYOUT_WIDE64, aos32=rt.LAYOUT_AOS32)[layout]
    out, st = gpu_ctx.render(U, rt.render_opts(mode=mid, gi_samples=8, layout=lay, flags=rt.FLAG_COUNT_WORK | rt.FLAG_TEXTURED))
    ref, rst = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mid, gi_samples=8, n_threads=8, surface=surf)
    assert np.array_equal(out["hit_tri"], ref["hit_tri"]) and np.array_equal(out["hit_t"], ref["hit_t"])
    d = np.abs(out["rgba8"].astype(np.int32) - ref["rgba8"].astype(np.int32))
    assert d.max() <= 1, f"RGB max diff {d.max()} LSB"
    assert np.array_equal(out["rgba8"], ref["rgba8"]), "specified to be bit-identical to the oracle"
    for k in COUNT_KEYS:
        assert st[k] == rst[k]
    plain, _ = gpu_ctx.render(U, rt.render_opts(mode=mid, gi_samples=8, layout=lay))
    hit = ref["hit_tri"] != 0xFFFFFFFF
    assert hit.mean() > 0.2 and (plain["rgba8"][hit] != out["rgba8"][hit]).any()          # the texture really is applied


def test_standin_s70k_textured_and_late_surface(gpu_ctx, oracle, rt):
    """cfg-3 style stand-in with three materials (textured / plain Kd / tinted second texture); the surface is bound
    AFTER the build this time (rt_set_surface reorders immediately)."""
    from raytracer_amd import scenes
    m = rt.normalize_mesh(scenes.s70k())
    gpu_ctx.set_node_encoding(1)
    gpu_ctx.set_mesh(m)
    gpu_ctx.build_lbvh()
    U = rt.make_uniforms(480, 270, cam_pos=(0.0, 0.0, 1.3))
    with pytest.raises(rt.RtError):
        gpu_ctx.render(U, rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED))          # no surface yet
    scenes.bind_standin_surface(gpu_ctx, m, tex_size=512)
    b = oracle.build_lbvh(m.positions, m.normals, m.vert_idx, m.norm_idx)
    tri_uv = oracle.gather_tri_uv(m.texcoords, m.tex_idx, m.material_ids, b["order"])
    assert np.array_equal(gpu_ctx.tri_uv().view(np.uint32), tri_uv.view(np.uint32))
    surf = oracle.Surface(tri_uv, m.materials, [scenes.procedural_texture(512, 0), scenes.procedural_texture(256, 1)])
    out, _ = gpu_ctx.render(U, rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED))
    ref, _ = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=1, n_threads=8, surface=surf)
    for k in ("hit_tri", "hit_t", "rgba8"):
        assert np.array_equal(out[k], ref[k]), k
    assert len(np.unique(ref["rgba8"].reshape(-1, 4), axis=0)) > 500


def test_imported_bvh_with_tri_uv_and_errors(gpu_ctx, oracle, rt, plane_textured):
    from raytracer_amd import scenes
    m = plane_textured
    s = oracle.build_sah(m.positions, m.normals, m.vert_idx, m.norm_idx, leaf=2)          # the reference's own tree
    tri_uv = oracle.gather_tri_uv(m.texcoords, m.tex_idx, m.material_ids, s["order"])
    gpu_ctx.set_node_encoding(1)
    gpu_ctx.set_bvh(s["nodes"], s["tri_pos"], s["tri_nor"])
    U = rt.make_uniforms(128, 96, **PLANE_VIEW)
    opts = rt.render_opts(mode=rt.MODE_SHADOW, flags=rt.FLAG_TEXTURED)
    with pytest.raises(rt.RtError):
        gpu_ctx.render(U, opts)                                   # an import drops the previous BVH's surface records
    with pytest.raises(rt.RtError):
        gpu_ctx.set_tri_uv(tri_uv[:-1])                           # one record per triangle
    gpu_ctx.set_tri_uv(tri_uv)
    mats = [((1.0, 1.0, 1.0), 3), ((0.3, 0.6, 0.9), -1)]
    gpu_ctx.set_materials(mats)
    tex = scenes.procedural_texture(128, 1)
    gpu_ctx.set_texture(0, scenes.procedural_texture(16, 0))
    ctx2 = rt.Context(0)
    try:
        ctx2.set_bvh(s["nodes"], s["tri_pos"], s["tri_nor"])
        ctx2.set_tri_uv(tri_uv)
        ctx2.set_materials(mats)
        with pytest.raises(rt.RtError):
            ctx2.render(U, opts)                                  # material 0 names slot 3, never set
        ctx2.set_texture(3, tex)
        out, _ = ctx2.render(U, opts)
    finally:
        ctx2.close()
    surf = oracle.Surface(tri_uv, mats, [tex] * 4)
    ref, _ = oracle.render(s["nodes"], s["tri_pos"], s["tri_nor"], U, mode=1, n_threads=8, surface=surf)
    for k in ("hit_tri", "hit_t", "rgba8"):
        assert np.array_equal(out[k], ref[k]), k
