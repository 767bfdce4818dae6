"""GPU: the reference's OWN tree — oracle restatement of buildLBVH (web/script.js:285-513, SAH) + flattenForSSBO —
imported through rt_set_bvh (the makeBuf uploads of script.js:743-748) and traversed by both layouts must give the
oracle's frame on that tree bit for bit, including leaves with more than one triangle (leaf > 1)."""
import numpy as np
import pytest

from conftest import random_soup

pytestmark = pytest.mark.gpu
COUNT_KEYS = ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests", "hit_updates", "pixels", "stack_drops")
PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))


@pytest.mark.parametrize("leaf", [1, 4])
@pytest.mark.parametrize("layout", ["wide64", "aos32"])
def test_sah_tree_import_plane(gpu_ctx, oracle, rt, plane_mesh, leaf, layout):
    m = plane_mesh
    s = oracle.build_sah(m.positions, m.normals, m.vert_idx, m.norm_idx, leaf=leaf)
    gpu_ctx.set_node_encoding(rt.NODES_V1)
    gpu_ctx.set_bvh(s["nodes"], s["tri_pos"], s["tri_nor"])
    lay = dict(wide64=rt.LAYOUT_WIDE64, aos32=rt.LAYOUT_AOS32)[layout]
    for mode, (w, h) in ((1, (160, 96)), (2, (64, 48))):
        U = rt.make_uniforms(w, h, **PLANE_VIEW)
        out, st = gpu_ctx.render(U, rt.render_opts(mode=mode, gi_samples=8, layout=lay, flags=rt.FLAG_COUNT_WORK))
        ref, rst = oracle.render(s["nodes"], s["tri_pos"], s["tri_nor"], U, mode=mode, gi_samples=8, n_threads=8)
        for k in ("hit_tri", "hit_t", "rgba8"):
            assert np.array_equal(out[k], ref[k]), f"leaf {leaf} {layout} mode {mode}: {k}"
        for k in COUNT_KEYS:
            assert st[k] == rst[k], f"{k}: gpu {st[k]} vs oracle {rst[k]}"
        assert (ref["hit_tri"] != 0xFFFFFFFF).mean() > 0.2


def test_sah_vs_lbvh_same_geometry(gpu_ctx, oracle, rt):
    """Closest-hit distance is a property of the geometry: GPU LBVH build and the imported SAH tree give the same t image
    (no stack drops at max_stack 32), and the SAH tree costs fewer box tests (the quality comparison of SURVEY §8f-2)."""
    mesh = random_soup(20000, 33, scale=1.0, size=0.05)
    mesh.positions -= 0.5
    U = rt.make_uniforms(256, 192, cam_pos=(0, 0, 1.2))
    U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
    gpu_ctx.set_node_encoding(rt.NODES_V1)
    gpu_ctx.set_mesh(mesh)
    gpu_ctx.build_lbvh()
    a, sa = gpu_ctx.render(U, rt.render_opts(mode=rt.MODE_PRIMARY, max_stack=32, flags=rt.FLAG_COUNT_WORK))
    s = oracle.build_sah(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    gpu_ctx.set_bvh(s["nodes"], s["tri_pos"], s["tri_nor"])
    b, sb = gpu_ctx.render(U, rt.render_opts(mode=rt.MODE_PRIMARY, max_stack=32, flags=rt.FLAG_COUNT_WORK))
    assert sa["stack_drops"] == 0 and sb["stack_drops"] == 0
    assert np.array_equal(a["hit_t"], b["hit_t"])
    assert sb["box_tests"] < sa["box_tests"]
