"""GPU LBVH build vs the CPU oracle: bit-exact Morton codes, sort order, topology, flat buffers.
All calls go through the C ABI (raytracer_amd.Context -> librtcore.so)."""
import numpy as np
import pytest

from conftest import random_soup

pytestmark = pytest.mark.gpu


def _check_build(ctx, oracle, mesh):
    ctx.set_mesh(mesh)
    stats = ctx.build_lbvh()
    ref = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    T = mesh.n_tri
    assert stats["n_tri"] == T and stats["n_nodes"] == 2 * T - 1
    bmin, inv = ctx.bounds()
    rb, ri = oracle.centroid_bounds(mesh.positions, mesh.vert_idx)
    assert np.array_equal(bmin.view(np.uint32), rb.view(np.uint32)), "centroid bounds min"
    assert np.array_equal(inv.view(np.uint32), ri.view(np.uint32)), "centroid bounds invExtent"
    assert np.array_equal(ctx.morton(), ref["codes"]), "Morton codes"
    assert np.array_equal(ctx.tri_order(), ref["order"]), "sort order (stable)"
    flat = ctx.flatten_for_ssbo()
    # topology: the A/B slots (child / leaf indices) bit for bit
    assert np.array_equal(flat["bvhData"][:, [3, 7]].view(np.uint32), ref["nodes"][:, [3, 7]].view(np.uint32)), "topology"
    # boxes: equal as floats (min/max are exact; only the sign of a zero may differ)
    assert np.array_equal(flat["bvhData"], ref["nodes"]), "node boxes"
    assert np.array_equal(flat["triPos"].view(np.uint32), ref["tri_pos"].view(np.uint32)), "triPos"
    assert np.array_equal(flat["triNor"].view(np.uint32), ref["tri_nor"].view(np.uint32)), "triNor"
    return stats


def test_quad(gpu_ctx, oracle, rt):
    _check_build(gpu_ctx, oracle, rt.generate_test_scene())


def test_single_triangle(gpu_ctx, oracle, rt):
    m = rt.Mesh(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), np.float32([[0, 0, 1]]), np.uint32([0, 1, 2]), np.uint32([0, 0, 0]))
    _check_build(gpu_ctx, oracle, m)


def test_plane_obj(gpu_ctx, oracle, plane_mesh):
    _check_build(gpu_ctx, oracle, plane_mesh)


@pytest.mark.parametrize("n_tri,seed,quantum", [(3, 1, None), (64, 2, None), (65, 3, 2), (4096, 4, None), (4097, 5, 4),
                                                 (10000, 6, 8), (100003, 7, None), (50000, 8, 1)])
def test_random_soups(gpu_ctx, oracle, n_tri, seed, quantum):
    # quantum snaps centroids to a coarse grid => many duplicate Morton codes => exercises the
    # stable-sort tie order and the index tie-break of the Karras delta (quantum=1: ALL codes equal-ish)
    _check_build(gpu_ctx, oracle, random_soup(n_tri, seed, quantum))


@pytest.mark.parametrize("n_tri", [255, 256, 257, 511, 512, 513, 1023, 1025, 65537])
def test_refit_block_boundaries(gpu_ctx, oracle, n_tri):
    """k_refit climbs inside 256-leaf blocks through LDS and across them through HBM: sizes around the block size."""
    _check_build(gpu_ctx, oracle, random_soup(n_tri, 100 + n_tri % 97, 3 if n_tri % 2 else None))


def test_degenerate_geometry(gpu_ctx, oracle, rt):
    """Zero-area and coincident triangles, a flat (zero-extent) axis, signed zeros, huge and denormal coordinates."""
    rng = np.random.default_rng(12)
    pts = rng.random((600, 3)).astype(np.float32)
    pts[:, 2] = 0.0                                    # flat in z: invExtent.z = 0
    pts[::7, 0] = -0.0
    pts[5::11] *= np.float32(1e-41)                    # denormals are kept (no flush to zero)
    idx = rng.integers(0, 600, (1500, 3)).astype(np.uint32)
    idx[:200, 1] = idx[:200, 0]                        # zero-area triangles
    idx[200:400] = idx[200]                            # 200 copies of one triangle: equal codes, index tie-break all the way
    nor = np.float32([[0, 0, 1]])
    m = rt.Mesh(pts, nor, idx.reshape(-1), np.zeros(4500, np.uint32))
    _check_build(gpu_ctx, oracle, m)
    big = rt.Mesh((pts * np.float32(1e30)).astype(np.float32), nor, idx.reshape(-1), np.zeros(4500, np.uint32))
    _check_build(gpu_ctx, oracle, big)
    same = rt.Mesh(np.float32([[1, 2, 3]] * 3), nor, np.zeros(3 * 300, np.uint32), np.zeros(900, np.uint32))   # every centroid identical
    _check_build(gpu_ctx, oracle, same)


def test_s70k(gpu_ctx, oracle):
    from raytracer_amd import scenes
    st = _check_build(gpu_ctx, oracle, scenes.s70k())
    assert st["ms_total"] > 0


def test_stage_times_are_opt_in(gpu_ctx, oracle):
    """rt_build_stats: ms_total is always first launch -> last; the stage fields only after rt_set_build_stage_timing(ctx, 1)
    (the events between the launches cost the build time), and the tree is the same either way."""
    from raytracer_amd import scenes
    mesh = scenes.s70k()
    gpu_ctx.set_mesh(mesh)
    st = gpu_ctx.build_lbvh()
    stages = ("ms_bounds", "ms_morton", "ms_sort", "ms_tree", "ms_refit")
    assert st["ms_total"] > 0 and all(st[k] == 0 for k in stages) and st["ms_pack"] == 0
    plain = gpu_ctx.flatten_for_ssbo()["bvhData"].copy()
    gpu_ctx.set_build_stage_timing(True)
    try:
        st2 = gpu_ctx.build_lbvh()
        assert all(st2[k] > 0 for k in stages)
        assert abs(st2["ms_total"] - sum(st2[k] for k in stages)) < 1e-4
        assert np.array_equal(gpu_ctx.flatten_for_ssbo()["bvhData"].view(np.uint32), plain.view(np.uint32))
    finally:
        gpu_ctx.set_build_stage_timing(False)


def test_sortedness_and_invariants_1m(gpu_ctx):
    """Full-size property checks (no oracle): sorted + stable, permutation, tree invariants, containment."""
    from raytracer_amd import scenes
    mesh = scenes.s1m()
    gpu_ctx.set_mesh(mesh)
    gpu_ctx.build_lbvh()
    codes, order = gpu_ctx.morton(), gpu_ctx.tri_order()
    T = mesh.n_tri
    sc = codes[order]
    assert np.all(sc[1:] >= sc[:-1]), "sorted"
    same = sc[1:] == sc[:-1]
    assert np.all(order[1:][same] > order[:-1][same]), "stable"
    assert np.array_equal(np.sort(order), np.arange(T, dtype=np.uint32)), "permutation"
    nodes = gpu_ctx.flatten_for_ssbo()["bvhData"]
    leaf = nodes[:, 7] < 0
    assert leaf.sum() == T and (~leaf).sum() == T - 1
    kids = nodes[~leaf][:, [3, 7]].astype(np.int64).reshape(-1)
    assert np.array_equal(np.sort(kids), np.arange(1, 2 * T - 1)), "every node but the root has exactly one parent"
    assert np.array_equal(np.sort(nodes[leaf][:, 3].astype(np.int64)), np.arange(T)), "leaves cover 0..T-1 once"
    c0, c1 = nodes[~leaf][:, 3].astype(np.int64), nodes[~leaf][:, 7].astype(np.int64)
    assert np.array_equal(nodes[~leaf][:, 0:3], np.minimum(nodes[c0][:, 0:3], nodes[c1][:, 0:3])), "box = union of children (min)"
    assert np.array_equal(nodes[~leaf][:, 4:7], np.maximum(nodes[c0][:, 4:7], nodes[c1][:, 4:7])), "box = union of children (max)"


SORT_TILE = 5120     # keys per onesweep tile (rt_build.hip: 256 threads x RT_SORT_KPT 20)


@pytest.mark.parametrize("n_tri,quantum", [(SORT_TILE, None), (SORT_TILE + 1, 3), (32 * SORT_TILE - 1, None), (32 * SORT_TILE, 5),
                                           (32 * SORT_TILE + 1, None), (33 * SORT_TILE + 77, 2), (65 * SORT_TILE + 5, None),
                                           (512 * SORT_TILE, 40), (512 * SORT_TILE + 1, None), (529 * SORT_TILE - 3, 7)])
def test_sort_tile_and_group_boundaries(gpu_ctx, rt, n_tri, quantum):
    """The sort's look-back has three regimes - one group (<= 32 tiles), groups of 32 (<= 512 tiles), groups of 16 (more) - and
    its words come in two widths (u16 per tile, u32 per group): key counts right at the tile and group boundaries, with and
    without heavy ties.  No oracle at these sizes: the order must be numpy's stable argsort of the device's own codes (which
    the smaller cases pin to the oracle), and every node must have exactly one parent."""
    mesh = random_soup(n_tri, 4000 + n_tri % 1009, quantum)
    gpu_ctx.set_mesh(mesh)
    st = gpu_ctx.build_lbvh()
    assert st["n_tri"] == n_tri and st["refit_fallbacks"] == 0
    codes, order = gpu_ctx.morton(), gpu_ctx.tri_order()
    assert np.array_equal(order, np.argsort(codes, kind="stable").astype(np.uint32)), "stable sort order"
    nodes = gpu_ctx.flatten_for_ssbo()["bvhData"]
    leaf = nodes[:, 7] < 0
    kids = nodes[~leaf][:, [3, 7]].astype(np.int64).reshape(-1)
    assert leaf.sum() == n_tri and np.array_equal(np.sort(kids), np.arange(1, 2 * n_tri - 1)), "every node but the root has exactly one parent"


def test_rebuild_is_deterministic(gpu_ctx, plane_mesh):
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    a = gpu_ctx.flatten_for_ssbo()["bvhData"].copy()
    gpu_ctx.build_lbvh()
    assert np.array_equal(a.view(np.uint32), gpu_ctx.flatten_for_ssbo()["bvhData"].view(np.uint32))


def test_errors(gpu_ctx, rt):
    with pytest.raises(rt.RtError):
        bad = rt.Mesh(np.float32([[0, 0, 0]]), np.float32([[0, 0, 1]]), np.uint32([0, 1, 2]), np.uint32([0, 0, 0]))
        gpu_ctx.set_mesh(bad)     # index out of range must be rejected on the host, never reach the GPU
    for poison in (np.nan, np.inf, -np.inf):
        with pytest.raises(rt.RtError):
            gpu_ctx.set_mesh(rt.Mesh(np.float32([[0, 0, 0], [1, poison, 0], [0, 1, 0]]), np.float32([[0, 0, 1]]), np.uint32([0, 1, 2]),
                                     np.uint32([0, 0, 0])))
    with pytest.raises(rt.RtError):
        gpu_ctx.build_lbvh() if False else gpu_ctx.set_mesh(rt.Mesh(np.zeros((0, 3), np.float32), np.float32([[0, 0, 1]]), np.zeros(0, np.uint32),
                                                                    np.zeros(0, np.uint32)))   # empty mesh
