"""Hand-derived known answers that pin the CPU oracle.

The reference has no tests or golden vectors for this path, so parity is UNPINNED BY THE
REFERENCE; these vectors are derived by hand from the WGSL text (cited per test).
"""
import math

import numpy as np


def test_expand_bits(oracle):                       # web/morton_compute.wgsl:18-25
    assert oracle.expand_bits(0x3FF) == 0x09249249
    assert oracle.expand_bits(1) == 1
    assert oracle.expand_bits(0) == 0
    assert oracle.expand_bits(0b10) == 0b1000
    assert oracle.expand_bits(0x7FF) == 0x09249249  # masked to 10 bits first
    for v in range(1024):                           # bit i of v lands on bit 3i
        e = oracle.expand_bits(v)
        assert e == sum(((v >> i) & 1) << (3 * i) for i in range(10))


def test_quantize(oracle):                          # web/morton_compute.wgsl:28-31
    assert oracle.quantize(0.5) == 512
    assert oracle.quantize(1.0) == 1023
    assert oracle.quantize(-0.1) == 0
    assert oracle.quantize(0.0) == 0
    assert oracle.quantize(0.99999) == 1023
    assert oracle.quantize(1.0 / 1024.0) == 1
    assert oracle.quantize(float("nan")) == 0      # pinned: NaN -> 0
    assert oracle.quantize(float("inf")) == 1023


def test_morton_interleave(oracle):                 # web/morton_compute.wgsl:68 — x is the most significant
    # one degenerate triangle per probe; bounds chosen so that nrm = centroid exactly
    def code(c):
        pos = np.float32([c, c, c])
        return int(oracle.morton(pos, np.uint32([0, 1, 2]), np.float32([0, 0, 0]), np.float32([1, 1, 1]))[0])
    assert code([1.0, 1.0, 1.0]) == 0x3FFFFFFF
    assert code([1.0, 0.0, 0.0]) == 0x24924924
    assert code([0.0, 1.0, 0.0]) == 0x12492492
    assert code([0.0, 0.0, 1.0]) == 0x09249249
    assert code([0.0, 0.0, 0.0]) == 0


def test_hash_and_rand2(oracle):                    # web/compute.wgsl:135-149
    assert oracle.hash_u32(0) == 0xC0A9496A
    # by hand: (0^61)^(0>>16)=61; 61+61*8=549; 549^(549>>4)=549^34=519; *0x27d4eb2d; ^>>15
    x = (519 * 0x27D4EB2D) & 0xFFFFFFFF
    assert oracle.hash_u32(0) == (x ^ (x >> 15))
    assert np.float32(0xC0A9496A) == np.float32(3232319744.0)
    # f32->u32 saturates: seed.x*43758+24521 >= 2^32 -> 0xFFFFFFFF
    a = oracle.rand2(3.0e9, 0.37)
    b = oracle.rand2(1.0e12, 0.37)
    assert a == b
    yi = int(np.float32(np.float32(0.37) * np.float32(199999.0)) + np.float32(1423.0))
    mix = 0xFFFFFFFF ^ (yi >> 6) ^ ((yi << 11) & 0xFFFFFFFF)
    assert a == float(np.float32(np.float32(oracle.hash_u32(mix)) * np.float32(1.0 / 4294967296.0)))
    assert 0.0 <= oracle.rand2(0.0, 0.0) < 1.0
    assert oracle.rand2(-5.0, -5.0) == oracle.rand2(-1.0, -1.0)   # negatives saturate to 0


def test_transcendental_kernels(oracle):
    """Pinned sin/cos/pow kernels: accuracy against libm (they need only be faithful)."""
    worst = 0.0
    for th in np.linspace(0.0, 6.5, 4001, dtype=np.float32):
        s, c = oracle.sincos(float(th))
        worst = max(worst, abs(s - math.sin(float(th))), abs(c - math.cos(float(th))))
    assert worst < 2.5e-7
    assert oracle.sincos(0.0) == (0.0, 1.0)
    for x in np.linspace(1e-6, 4.0, 2001, dtype=np.float32):
        assert abs(oracle.pow_(float(x), 1 / 2.2) - float(x) ** (1 / 2.2)) <= 4e-6 * float(x) ** (1 / 2.2)
    assert oracle.pow_(0.0, 1 / 2.2) == 0.0
    assert oracle.pow_(-1.0, 1 / 2.2) == 0.0
    assert oracle.pow_(1.0, 1 / 2.2) == 1.0


def test_quad_tree_and_bounds(oracle, rt):          # generateTestScene, web/script.js:639-675
    m = rt.generate_test_scene()
    bmin, inv = oracle.centroid_bounds(m.positions, m.vert_idx)
    # centroids (1/3,-1/3,0) and (-1/3,1/3,0): zero z extent => invExtent.z = 0
    third = np.float32(np.float32(np.float32(-1 + 1) + np.float32(1)) / np.float32(3))
    assert bmin[0] == -third and bmin[1] == -third and bmin[2] == 0
    assert inv[2] == 0.0 and inv[0] == np.float32(1) / np.float32(third + third)
    b = oracle.build_lbvh(m.positions, m.normals, m.vert_idx, m.norm_idx)
    assert list(b["codes"]) == [0x24924924, 0x12492492]        # tri0: x=1023,y=0 ; tri1: x=0,y=1023
    assert list(b["order"]) == [1, 0]
    n = b["nodes"]
    assert n.shape == (3, 8)
    assert list(n[0]) == [-1, -1, 0, 1, 1, 1, 0, 2]             # root: children 1,2; box [-1,1]^2 x {0}
    assert list(n[1]) == [-1, -1, 0, 0, 1, 1, 0, -1]            # leaf: tri_start 0, -count
    assert list(n[2]) == [-1, -1, 0, 1, 1, 1, 0, -1]
    assert np.array_equal(b["tri_pos"][0, :, :3], m.positions[[0, 2, 3]])   # sorted tri 0 = input tri 1
    assert np.all(b["tri_pos"][:, :, 3] == 0)


def test_default_uniforms(oracle):                  # web/script.js:880-929, web/index.html:12-22
    U = oracle.make_uniforms(width=1920, height=1080)
    fl = 1 / math.tan(math.pi / 6)
    assert list(U[0:4]) == [0, 0, 5, 0]
    assert [float(x) for x in U[4:8]] == [5.0, -5.0, 5.0, float(np.float32(0.3))]
    assert U[8] == np.float32(fl * 1920 / 1080) and U[9] == 0 and U[10] == 0
    assert U[13] == np.float32(fl) and U[18] == np.float32(-fl)
    M = U[20:36].reshape(4, 4)
    assert M[0, 0] == 1 and M[1, 1] == -1 and M[2, 2] == -1 and M[3, 3] == 1   # rotX - pi (script.js:148)
    assert U[36] == 1920 and U[37] == 1080 and U[38] == 0 and np.all(U[39:] == 0)


def test_quad_center_pixel(oracle, rt):             # SURVEY §8c quad known answers
    m = rt.generate_test_scene()
    b = oracle.build_lbvh(m.positions, m.normals, m.vert_idx, m.norm_idx)
    U = oracle.make_uniforms(width=65, height=65)   # odd size: pixel (32,32) is the exact centre, sx = sy = 0
    out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=oracle.MODE_PT)
    # default sliders: model ~ diag(1,-1,-1): the ray meets the quad from behind => culled => miss
    assert list(out["rgba8"][32, 32]) == [0, 0, 0, 255]
    assert out["hit_tri"][32, 32] == 0xFFFFFFFF
    assert st["rays_shadow"] == 0 and st["rays_bounce"] == 0
    U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
    out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=oracle.MODE_PT)
    assert out["hit_t"][32, 32] == 5.0              # ro=(0,0,5), rd=(0,0,-1), quad at z=0
    assert out["hit_tri"][32, 32] in (0, 1)
    # centre ray hits the shared diagonal: the FIRST triangle in traversal order keeps it (strict <)
    # root children: node1 (sorted tri 0) is tested first when b0.x <= b1.x (equal boxes) => tri 0
    assert out["hit_tri"][32, 32] == 0
    # direct term: n=(0,0,1), L = normalize((5,-5,5)-(0,0,0)) => diff = 1/sqrt(3); light unoccluded
    r, g, bl, a = out["rgba8"][32, 32]
    assert a == 255 and bl == 0 and r > g > 0
    assert abs(int(r) - round(255 * (1 / math.sqrt(3)) ** (1 / 2.2))) <= 60   # + indirect (quad sees no other geometry => 0)
    assert int(r) == round(255 * (1 / math.sqrt(3)) ** (1 / 2.2))
