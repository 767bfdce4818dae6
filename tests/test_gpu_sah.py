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


def _soup_view(rt):
    U = rt.make_uniforms(96, 64, cam_pos=(0, 0, 1.2))
    U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
    return U


def _check_both_layouts(gpu_ctx, oracle, rt, nodes, tp, tn, U, what):
    res = {}
    for name, lay in (("wide64", rt.LAYOUT_WIDE64), ("aos32", rt.LAYOUT_AOS32)):
        out, st = gpu_ctx.render(U, rt.render_opts(mode=rt.MODE_SHADOW, layout=lay, flags=rt.FLAG_COUNT_WORK))
        ref, rst = oracle.render(nodes, tp, tn, U, mode=1, n_threads=8)
        for k in ("hit_tri", "hit_t", "rgba8"):
            assert np.array_equal(out[k], ref[k]), f"{what} {name}: {k}"
        for k in COUNT_KEYS:
            assert st[k] == rst[k], f"{what} {name} {k}: gpu {st[k]} vs oracle {rst[k]}"
        res[name] = out
    for k in ("hit_tri", "hit_t", "rgba8"):
        assert np.array_equal(res["wide64"][k], res["aos32"][k])
    return res["wide64"]


def test_single_leaf_of_256_triangles(gpu_ctx, oracle, rt):
    """One root leaf holding 256 triangles (N = 1): the walk's step budget must cover a step per leaf TRIANGLE, not only
    per node (a budget of n_nodes + 64 tested 64 of the 256 triangles and silently dropped the hits behind them)."""
    mesh = random_soup(256, 5, size=0.2)
    mesh.positions -= 0.5
    tp = np.zeros((256, 3, 4), np.float32)
    tp[:, :, :3] = mesh.positions[mesh.vert_idx].reshape(256, 3, 3)
    tn = np.zeros((256, 3, 4), np.float32)
    tn[:, :, :3] = mesh.normals[mesh.norm_idx].reshape(256, 3, 3)
    lo, hi = tp[:, :, :3].reshape(-1, 3).min(0), tp[:, :, :3].reshape(-1, 3).max(0)
    nodes = np.float32([[lo[0], lo[1], lo[2], 0.0, hi[0], hi[1], hi[2], -256.0]])
    gpu_ctx.set_node_encoding(rt.NODES_V1)
    gpu_ctx.set_bvh(nodes, tp, tn)
    out = _check_both_layouts(gpu_ctx, oracle, rt, nodes, tp, tn, _soup_view(rt), "256-triangle leaf")
    hit = out["hit_tri"][out["hit_tri"] != 0xFFFFFFFF]
    assert hit.size > 200 and hit.max() >= 200           # triangles from the back of the leaf win pixels too


@pytest.mark.parametrize("leaf", [8, 64])
def test_sah_tree_with_big_leaves_on_a_soup(gpu_ctx, oracle, rt, leaf, tmp_path):
    mesh = random_soup(3000, 17, size=0.08)
    mesh.positions -= 0.5
    s = oracle.build_sah(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, leaf=leaf)
    assert s["nodes"].shape[0] != 2 * 3000 - 1
    gpu_ctx.set_node_encoding(rt.NODES_V1)
    gpu_ctx.set_bvh(s["nodes"], s["tri_pos"], s["tri_nor"])
    U = _soup_view(rt)
    out = _check_both_layouts(gpu_ctx, oracle, rt, s["nodes"], s["tri_pos"], s["tri_nor"], U, f"leaf {leaf}")
    # the cache file of a tree with N != 2T-1 carries a trailer with N and T and loads back into the same frame
    path = tmp_path / f"sah_leaf{leaf}.bvh"
    gpu_ctx.save_bvh(path)
    assert path.stat().st_size == 32 * s["nodes"].shape[0] + 96 * 3000 + 16
    c = rt.Context(0)
    try:
        c.load_bvh(path)
        assert c.counts() == (3000, s["nodes"].shape[0])
        again, _ = c.render(U, rt.render_opts(mode=rt.MODE_SHADOW))
        assert np.array_equal(again["rgba8"], out["rgba8"]) and np.array_equal(again["hit_tri"], out["hit_tri"])
    finally:
        c.close()


def test_adopted_device_bvh_with_a_leaf_past_the_triangle_buffer_is_refused(gpu_ctx, rt, plane_mesh):
    """rt_bvh_adopt_device never sees the host-side validation: the pack kernel itself must refuse a leaf whose
    [start, start + count) leaves the triangle buffer (k_render would read tri_pre / tri_nor out of bounds)."""
    import torch
    gpu_ctx.set_node_encoding(rt.NODES_V1)
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    f = gpu_ctx.flatten_for_ssbo()
    T, N = gpu_ctx.counts()
    bad = f["bvhData"].copy()
    bad[N - 1, 3] = float(T - 1)                          # last leaf: start = T - 1 ...
    bad[N - 1, 7] = -3.0                                  # ... count 3: two triangles past the end
    dev = torch.device("cuda:0")
    dn, dp, dq = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (bad, f["triPos"], f["triNor"]))
    c = rt.Context(0)
    try:
        with pytest.raises(rt.RtError) as e:
            c.adopt_bvh_device(dn.data_ptr(), dp.data_ptr(), dq.data_ptr(), N, T)
        assert e.value.code == -1
        good = torch.from_numpy(np.ascontiguousarray(f["bvhData"])).to(dev)
        c.adopt_bvh_device(good.data_ptr(), dp.data_ptr(), dq.data_ptr(), N, T)
    finally:
        c.close()
