"""GPU: the reference's one fully textured asset rendered with its OWN material file and images - web/plane.obj +
models/plane_material.mtl + Material_85_baseColor.jpg / Material_86_baseColor.png - through rt.load_obj ->
rt.bind_obj_surface -> RT_FLAG_TEXTURED, against the oracle, bit for bit.  The images are the committed fixtures of
tests/golden/make_textures.py (decoded once with PIL where /root/reference exists, box-filtered 2048^2 -> 512^2, binary
PPM); they are written under the names the MTL's map_Kd lines ask for (load_texture goes by the file's magic, not its
suffix).  The reference itself never samples a texture: parity unpinned by the reference, pinned by
tests/test_oracle_texture.py's hand-derived answers."""
import gzip
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
COUNT_KEYS = ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests", "hit_updates", "pixels", "stack_drops")
IMAGES = {"Material_85_baseColor.jpg": "Material_85_baseColor_512.ppm.gz", "Material_86_baseColor.png": "Material_86_baseColor_512.ppm.gz"}


@pytest.fixture(scope="module")
def plane_asset(rt, tmp_path_factory):
    d = tmp_path_factory.mktemp("plane_asset")
    with gzip.open(os.path.join(GOLDEN, "plane.obj.gz"), "rb") as f:
        (d / "plane.obj").write_bytes(f.read())
    shutil.copy(os.path.join(GOLDEN, "plane_material.mtl"), d / "plane_material.mtl")     # the file plane.obj's mtllib names
    for name, fixture in IMAGES.items():
        with gzip.open(os.path.join(GOLDEN, fixture), "rb") as f:
            (d / name).write_bytes(f.read())
    mesh = rt.normalize_mesh(rt.load_obj(d / "plane.obj"))
    return mesh, str(d)


def test_plane_obj_with_its_own_mtl_and_images(gpu_ctx, oracle, rt, plane_asset):
    mesh, d = plane_asset
    assert [m["name"] for m in mesh.materials] == ["Material_85", "Material_86"]
    assert set(np.unique(mesh.material_ids)) == {0, 1} and mesh.texcoords.shape[0] >= 6088
    gpu_ctx.set_node_encoding(1)
    gpu_ctx.set_mesh(mesh)
    bound = rt.bind_obj_surface(gpu_ctx, mesh, d)
    assert bound == [("Material_85_baseColor.jpg", 0), ("Material_86_baseColor.png", 1)]
    gpu_ctx.build_lbvh()
    texs = [rt.load_texture(os.path.join(d, n)) for n, _ in bound]
    assert all(t.shape == (512, 512, 3) for t in texs) and not np.array_equal(texs[0], texs[1])
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    tri_uv = oracle.gather_tri_uv(mesh.texcoords, mesh.tex_idx, mesh.material_ids, b["order"])
    assert np.array_equal(gpu_ctx.tri_uv().view(np.uint32), tri_uv.view(np.uint32))
    surf = oracle.Surface(tri_uv, [((1.0, 1.0, 1.0), 0), ((1.0, 1.0, 1.0), 1)], texs)     # Kd 1 1 1 in both materials (test_host.py)
    for mode, (w, h), spp in ((rt.MODE_SHADOW, (320, 192), 32), (rt.MODE_PT, (96, 64), 8)):
        U = rt.make_uniforms(w, h, **PLANE_VIEW)
        out, st = gpu_ctx.render(U, rt.render_opts(mode=mode, gi_samples=spp, flags=rt.FLAG_COUNT_WORK | rt.FLAG_TEXTURED))
        ref, rst = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, gi_samples=spp, n_threads=8, surface=surf)
        assert np.array_equal(out["hit_tri"], ref["hit_tri"]) and np.array_equal(out["hit_t"], ref["hit_t"])
        d8 = np.abs(out["rgba8"].astype(np.int32) - ref["rgba8"].astype(np.int32))
        assert d8.max() <= 1, f"RGB max diff {d8.max()} LSB"                              # north_star's tolerance ...
        assert np.array_equal(out["rgba8"], ref["rgba8"])                                 # ... and the stronger claim: identical
        for k in COUNT_KEYS:
            assert st[k] == rst[k], k
        hit = ref["hit_tri"] != 0xFFFFFFFF
        plain, _ = gpu_ctx.render(U, rt.render_opts(mode=mode, gi_samples=spp))
        assert hit.mean() > 0.2 and (plain["rgba8"][hit] != out["rgba8"][hit]).mean() > 0.5   # the photographs really are sampled
        assert len(np.unique(out["rgba8"][hit].reshape(-1, 4), axis=0)) > 1000
