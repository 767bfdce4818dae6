"""CPU-only: the oracle's restatement of the reference's OWN builder (web/script.js:285-513 `buildLBVH`, a recursive
20-bin SAH builder) + flattenForSSBO (script.js:526-561).  SURVEY §8f-2.  Parity unpinned by the reference (it holds no
builder fixtures, and running script.js is denied): pinned by hand-derived answers and structural properties."""
import numpy as np
import pytest

from conftest import random_soup


def _walk(nodes):
    """DFS pre-order walk; returns per-node (lo, hi) triangle ranges and the visiting order."""
    N = nodes.shape[0]
    rng = np.full((N, 2), -1, np.int64)
    order = []
    stack = [(0, False)]
    while stack:
        n, done = stack.pop()
        a, b = nodes[n, 3], nodes[n, 7]
        if b < 0:
            order.append(n)
            rng[n] = (int(a), int(a) - int(b) - 1)
            continue
        l, r = int(a), int(b)
        if done:
            rng[n] = (rng[l, 0], rng[r, 1])
        else:
            order.append(n)
            stack.append((n, True))
            stack.append((r, False))
            stack.append((l, False))
    return rng, order


@pytest.mark.parametrize("n_tri,seed,leaf", [(1, 0, 1), (2, 1, 1), (3, 2, 1), (300, 3, 1), (5000, 4, 1), (5000, 5, 4), (800, 6, 8)])
def test_sah_invariants(oracle, n_tri, seed, leaf):
    mesh = random_soup(n_tri, seed)
    b = oracle.build_sah(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx, leaf=leaf)
    nodes, order, T = b["nodes"], b["order"], n_tri
    N = nodes.shape[0]
    if leaf == 1:
        assert N == 2 * T - 1
    assert np.array_equal(np.sort(order), np.arange(T, dtype=np.uint32))          # a permutation
    tri = mesh.positions[mesh.vert_idx.reshape(-1, 3)]
    assert np.array_equal(b["tri_pos"][:, :, :3], tri[order]) and np.all(b["tri_pos"][:, :, 3] == 0)
    assert np.array_equal(b["tri_nor"][:, :, :3], mesh.normals[mesh.norm_idx.reshape(-1, 3)][order])
    rng, visit = _walk(nodes)
    # script.js:331-332: nodes are numbered in DFS pre-order, so the left child is always parent + 1
    assert visit == list(range(N))
    inner = nodes[:, 7] >= 0
    assert np.all(nodes[inner, 3] == np.nonzero(inner)[0] + 1)
    assert rng[0, 0] == 0 and rng[0, 1] == T - 1
    leaves = np.nonzero(~inner)[0]
    cnt = (-nodes[leaves, 7]).astype(np.int64)
    assert cnt.min() >= 1 and cnt.max() <= leaf and cnt.sum() == T
    # leaves in DFS order tile 0..T-1 exactly once; children split the parent's range
    assert np.array_equal(nodes[leaves, 3].astype(np.int64), np.concatenate([[0], np.cumsum(cnt)[:-1]]))
    for n in np.nonzero(inner)[0]:
        l, r = int(nodes[n, 3]), int(nodes[n, 7])
        assert rng[l, 1] + 1 == rng[r, 0]
    # every node's box is the exact union of its triangles' true bounds (script.js:336-344)
    tp = b["tri_pos"][:, :, :3]
    tmin, tmax = tp.min(1), tp.max(1)
    for n in range(0, N, max(1, N // 400)):
        lo, hi = rng[n]
        assert np.array_equal(nodes[n, 0:3], tmin[lo:hi + 1].min(0)) and np.array_equal(nodes[n, 4:7], tmax[lo:hi + 1].max(0))


def test_sah_quad_known_answer(oracle, rt):
    """generateTestScene (script.js:639-675): two triangles.  Centroids (1/3,-1/3,0) and (-1/3,1/3,0): x and y extents
    tie, `ext[0] > ext[1]` is false and `ext[1] > ext[2]` is true => axis 1 (script.js:364-365).  Triangle 0 (y = -1/3)
    lands in bin 0, triangle 1 in bin 19; the first candidate split (i = 0) already has nL = nR = 1 and both child
    areas are 0, so bestBin = 0, splitPos = cMn + ext/20, and the order stays (0, 1)."""
    m = rt.generate_test_scene()
    b = oracle.build_sah(m.positions, m.normals, m.vert_idx, m.norm_idx)
    assert np.array_equal(b["order"], [0, 1])
    n = b["nodes"]
    assert n.shape == (3, 8)
    assert list(n[0]) == [-1, -1, 0, 1, 1, 1, 0, 2]
    assert list(n[1, [3, 7]]) == [0, -1] and list(n[2, [3, 7]]) == [1, -1]


def test_sah_median_fallback_on_coincident_centroids(oracle, rt):
    """All centroids equal => ext = 0, invExt = 0, every triangle in bin 0, no candidate has nR > 0 => bestBin = -1 =>
    median split `start + (count >> 1)` (script.js:481-487)."""
    base = np.array([[-1, -1, 0], [1, -1, 0], [0, 2, 0]], np.float32)       # centroid (0,0,0)
    pos = np.concatenate([base * s for s in (1.0, 0.5, 0.25, 0.125, 2.0)]).astype(np.float32)
    idx = np.arange(15, dtype=np.uint32)
    nor = np.tile(np.array([[0, 0, 1]], np.float32), (15, 1))
    b = oracle.build_sah(pos, nor, idx, idx)
    assert np.array_equal(b["order"], np.arange(5))
    n = b["nodes"]
    # 5 -> (2, 3); 2 -> (1, 1); 3 -> (1, 2); 2 -> (1, 1): pre-order ids
    assert [int(x) for x in n[:, 3]] == [1, 2, 0, 1, 5, 2, 7, 3, 4]
    assert [int(x) for x in n[:, 7]] == [4, 3, -1, -1, 6, -1, 8, -1, -1]


def test_sah_tree_renders_like_lbvh_where_order_cannot_matter(oracle, rt):
    """Same scene through the two trees: the closest-hit distance is a property of the geometry, not of the tree, as
    long as nothing is dropped (max_stack 64) — t images must agree exactly; ids map through the two orders except
    at exact-t ties."""
    mesh = random_soup(3000, 21, scale=1.0, size=0.2)
    mesh.positions -= 0.5
    a = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    s = oracle.build_sah(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    U = rt.make_uniforms(64, 48, cam_pos=(0, 0, 1.6))
    U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
    ra, sa = oracle.render(a["nodes"], a["tri_pos"], a["tri_nor"], U, mode=0, max_stack=64)
    rs, ss = oracle.render(s["nodes"], s["tri_pos"], s["tri_nor"], U, mode=0, max_stack=64)
    assert sa["stack_drops"] == 0 and ss["stack_drops"] == 0
    assert np.array_equal(ra["hit_t"], rs["hit_t"])
    hit = ra["hit_tri"] != 0xFFFFFFFF
    assert hit.mean() > 0.1
    same = a["order"][ra["hit_tri"][hit]] == s["order"][rs["hit_tri"][hit]]
    assert same.mean() > 0.99
    # the quality the SAH buys: fewer box tests per ray than the Morton tree on the same frame
    assert ss["box_tests"] < sa["box_tests"]
