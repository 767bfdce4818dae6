"""CPU-only: structural properties of the oracle's LBVH (SURVEY.md §8c property tests) and an
independent numpy cross-check of its Morton / sort stages."""
import numpy as np
import pytest

from conftest import random_soup


def _np_morton(mesh):
    p = mesh.positions[mesh.vert_idx.reshape(-1, 3)]
    c = ((p[:, 0] + p[:, 1]) + p[:, 2]) / np.float32(3.0)
    mn, mx = c.min(0), c.max(0)
    ext = mx - mn
    inv = np.where(ext > 0, np.float32(1.0) / np.where(ext > 0, ext, 1).astype(np.float32), np.float32(0)).astype(np.float32)
    n = ((c - mn) * inv).astype(np.float32)
    q = np.clip(np.floor(n * np.float32(1024.0)), 0, 1023).astype(np.uint32)

    def expand(v):
        out = np.zeros_like(v)
        for i in range(10):
            out |= ((v >> i) & 1) << (3 * i)
        return out
    return (expand(q[:, 0]) << 2) | (expand(q[:, 1]) << 1) | expand(q[:, 2]), mn, inv


@pytest.mark.parametrize("n_tri,seed,quantum", [(1, 0, None), (2, 1, None), (3, 2, None), (257, 3, None), (5000, 4, 4), (20000, 5, None),
                                                 (3000, 6, 1)])
def test_lbvh_invariants(oracle, n_tri, seed, quantum):
    mesh = random_soup(n_tri, seed, quantum)
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    T = n_tri
    codes, order, nodes = b["codes"], b["order"], b["nodes"]
    ref_codes, mn, inv = _np_morton(mesh)
    assert np.array_equal(codes, ref_codes)
    bmin, binv = oracle.centroid_bounds(mesh.positions, mesh.vert_idx)
    assert np.array_equal(bmin, mn) and np.array_equal(binv, inv)
    assert np.array_equal(order, np.argsort(codes, kind="stable").astype(np.uint32))         # stable sort
    assert nodes.shape == (2 * T - 1, 8)
    leaf = nodes[:, 7] < 0
    assert leaf.sum() == T
    if T == 1:
        assert leaf[0]
        return
    assert not leaf[0]
    # Karras numbering: internal i -> i, leaf j -> (T-1)+j
    assert not leaf[:T - 1].any() and leaf[T - 1:].all()
    assert np.array_equal(nodes[T - 1:, 3], np.arange(T, dtype=np.float32)) and np.all(nodes[T - 1:, 7] == -1)
    kids = nodes[:T - 1][:, [3, 7]].astype(np.int64)
    assert np.array_equal(np.sort(kids.reshape(-1)), np.arange(1, 2 * T - 1))               # one parent each, root has none
    c0, c1 = kids[:, 0], kids[:, 1]
    assert np.array_equal(nodes[:T - 1, 0:3], np.minimum(nodes[c0, 0:3], nodes[c1, 0:3]))
    assert np.array_equal(nodes[:T - 1, 4:7], np.maximum(nodes[c0, 4:7], nodes[c1, 4:7]))
    # leaf boxes are the true triangle bounds of the BVH-ordered triangles
    tp = b["tri_pos"][:, :, :3]
    assert np.array_equal(nodes[T - 1:, 0:3], tp.min(1)) and np.array_equal(nodes[T - 1:, 4:7], tp.max(1))
    assert np.array_equal(tp, mesh.positions[mesh.vert_idx.reshape(-1, 3)[order]])
    # every internal node covers a contiguous range of sorted leaves, child0 to the left of child1
    lo = np.zeros(2 * T - 1, np.int64); hi = np.zeros(2 * T - 1, np.int64)
    lo[T - 1:] = hi[T - 1:] = np.arange(T)
    done = leaf.copy()
    for _ in range(200):
        ready = ~done[:T - 1] & done[c0] & done[c1]
        idx = np.nonzero(ready)[0]
        if idx.size == 0:
            break
        assert np.all(hi[c0[idx]] + 1 == lo[c1[idx]])
        lo[idx], hi[idx] = lo[c0[idx]], hi[c1[idx]]
        done[idx] = True
    assert done.all() and lo[0] == 0 and hi[0] == T - 1


def test_karras_matches_bruteforce_split(oracle):
    """Topology cross-check against a direct top-down construction over the augmented keys."""
    rng = np.random.default_rng(7)
    for T in (2, 3, 5, 16, 100, 777):
        keys = np.sort(rng.integers(0, 64 if T > 50 else 1 << 30, T).astype(np.uint32))
        c0, c1, par = oracle.karras(keys)
        aug = (keys.astype(np.uint64) << np.uint64(32)) | np.arange(T, dtype=np.uint64)
        exp0, exp1 = {}, {}

        def build(lo, hi):
            # split position: highest differing bit between first and last key
            x = int(aug[lo]) ^ int(aug[hi])
            bit = x.bit_length() - 1
            split = lo
            while ((int(aug[split + 1]) >> bit) & 1) == ((int(aug[lo]) >> bit) & 1):
                split += 1
            return split
        stack = [(0, T - 1)]
        while stack:
            lo, hi = stack.pop()
            g = build(lo, hi)
            # Karras: the internal node's index is the end of its range adjacent to the split's sibling boundary
            left = (T - 1) + g if lo == g else g
            right = (T - 1) + g + 1 if hi == g + 1 else g + 1
            node = lo if lo == 0 and hi == T - 1 else None
            exp0[(lo, hi)] = (g, left, right)
            if lo != g:
                stack.append((lo, g))
            if hi != g + 1:
                stack.append((g + 1, hi))
        # walk the oracle's tree and compare ranges + split points
        def rng_of(n):
            if n >= T - 1:
                return (n - (T - 1), n - (T - 1))
            a, b = rng_of(int(c0[n])), rng_of(int(c1[n]))
            return (a[0], b[1])
        for n in range(T - 1):
            lo, hi = rng_of(n)
            g, left, right = exp0[(lo, hi)]
            assert (int(c0[n]), int(c1[n])) == (left, right)
            assert n in (lo, hi)


def test_render_rows_and_threads_are_consistent(oracle, rt):
    mesh = random_soup(300, 11, scale=1.0, size=0.4)
    mesh.positions -= 0.5
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    U = rt.make_uniforms(40, 30, cam_pos=(0, 0, 1.5))
    a, sa = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=2, gi_samples=4, n_threads=1)
    c, sc = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=2, gi_samples=4, n_threads=5)
    for k in a:
        assert np.array_equal(a[k], c[k])
    for k in sa:
        if k != "seconds":
            assert sa[k] == sc[k]
    d, sd = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=2, gi_samples=4, rows=(3, 30, 7))
    rows = list(range(3, 30, 7))
    assert np.array_equal(d["rgba8"][rows], a["rgba8"][rows]) and sd["pixels"] == len(rows) * 40
    assert (a["hit_tri"] != 0xFFFFFFFF).any()


def _brute_force_closest(tri, ro, rd):
    """Front-face Moeller-Trumbore (compute.wgsl:107-132 predicate) of every ray against EVERY triangle, float64, no BVH."""
    v0, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    best_t = np.full(len(ro), np.inf)
    best_i = np.full(len(ro), -1, np.int64)
    for r in range(len(ro)):
        p = np.cross(rd[r], e2)
        det = np.einsum("ij,ij->i", e1, p)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = ro[r] - v0
            u = np.einsum("ij,ij->i", s, p) * inv
            q = np.cross(s, e1)
            v = (q @ rd[r]) * inv
            t = np.einsum("ij,ij->i", e2, q) * inv
        ok = (det > 1e-10) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-10)
        if ok.any():
            t = np.where(ok, t, np.inf)
            best_i[r] = int(np.argmin(t))
            best_t[r] = t[best_i[r]]
    return best_i, best_t


@pytest.mark.parametrize("scene", ["plane", "soup"])
def test_traversal_finds_the_brute_force_closest_hit(oracle, rt, plane_mesh, scene):
    """Ground truth that does not share the tree code: with a stack deep enough that nothing is dropped, the oracle's BVH
    traversal must return, for every pixel, the closest front-facing triangle of an exhaustive float64 test of all triangles."""
    if scene == "plane":
        mesh = plane_mesh
        U = rt.make_uniforms(48, 32, cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
    else:
        mesh = random_soup(3000, 31, scale=1.0, size=0.3)
        mesh.positions -= 0.5
        U = rt.make_uniforms(48, 32, cam_pos=(0.1, -0.05, 1.4), yaw=0.05)
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=0, max_stack=64)
    assert st["stack_drops"] == 0
    w, h = 48, 32
    # the rays of cs_main in float64 from the same 44 uniform floats (compute.wgsl:419-432, :351-358)
    Ud = U.astype(np.float64)
    M = Ud[20:36].reshape(4, 4).T                                   # column-major storage
    ys, xs = np.mgrid[0:h, 0:w]
    sx, sy = (xs + 0.5) / w * 2 - 1, (ys + 0.5) / h * 2 - 1
    rd_w = Ud[8:11] * sx[..., None] + Ud[12:15] * sy[..., None] + Ud[16:19]
    rd_w /= np.linalg.norm(rd_w, axis=-1, keepdims=True)
    ro = (M @ np.append(Ud[0:3], 1.0))[:3]
    rd = rd_w.reshape(-1, 3) @ M[:3, :3].T
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    tri = b["tri_pos"][:, :, :3].astype(np.float64)                 # BVH order = the ids the traversal reports
    bi, bt = _brute_force_closest(tri, np.tile(ro, (w * h, 1)), rd)
    got_i = out["hit_tri"].reshape(-1).astype(np.int64)
    got_i[got_i == 0xFFFFFFFF] = -1
    got_t = out["hit_t"].reshape(-1).astype(np.float64)
    hit = bi >= 0
    assert hit.mean() > 0.15
    # f32 vs f64 rays differ in the last bits: allow disagreement only where a second triangle is within 1e-5 of the winner (edges)
    same = got_i == bi
    close = np.abs(got_t - bt) <= 1e-5 * np.maximum(1.0, np.abs(bt))
    assert np.all((got_i >= 0) == hit) or np.mean((got_i >= 0) != hit) < 0.01      # silhouette pixels may flip
    both = hit & (got_i >= 0)
    assert np.all(close[both]), "hit distance differs from the exhaustive test"
    assert same[both].mean() > 0.98
