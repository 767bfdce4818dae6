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


def _derivation():
    import importlib.util
    import os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derive_known_answers.py")
    spec = importlib.util.spec_from_file_location("derive_known_answers", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_soft_shadow_pixel_worked_in_float64(oracle):
    """computeDirectRadiance_OS_fast (compute.wgsl:199-256) on a tilted receiver under a small occluder, worked in float64 by
    tests/golden/derive_known_answers.py (no oracle / library code in it): pixel (4, 3) of an 8 x 8 frame hits the receiver at
    t = 4.92555...; of the four stratified directions (cells 0..3 of the 6 x 6 grid) all are `visible` (cos to the light
    0.831, 0.827, 0.868, 0.591 > 0.5), samples 1 and 2 hit the occluder, 0 and 3 pass it: occ = 2, shadowFactor = 0.5,
    direct = 0.7824 * 0.5, colour = pow((1, 0.84, 0) * 0.3912, 1/2.2) -> (166, 154, 0, 255)."""
    d = _derivation()
    a = d.case_a()
    assert a["hit_tri"] == 1 and a["occ"] == 2 and a["shadow_factor"] == 0.5
    assert [(s["visible"], s["hits_occluder"]) for s in a["samples"]] == [(True, False), (True, True), (True, True), (True, False)]
    assert min(abs(s["cos_to_light"] - 0.5) for s in a["samples"]) > 0.05            # no flag sits on its threshold
    assert a["rgba8"] == [166, 154, 0, 255]
    nodes, tp, tn = d.flat_buffers()
    out, _ = oracle.render(nodes, tp, tn, d.uniforms(d.LIGHT_A), mode=oracle.MODE_SHADOW, gi_samples=32, soft_samples=4)
    x, y = a["pixel"]
    assert out["hit_tri"][y, x] == 1
    assert abs(float(out["hit_t"][y, x]) - a["hit_t"]) < 2e-6 * a["hit_t"]
    assert list(out["rgba8"][y, x]) == a["rgba8"]
    # the jitters are the PRNG's: sample 0's first one by hand - seed.x = f32(28 + 17) + f32(hash_u32(0)) = 3232319744 (45 is
    # below half an ulp of 256), so xi saturates to 0xFFFFFFFF; seed.y = 0.37: yi = u32(0.37f * 199999 + 1423) = 75422
    mix = 0xFFFFFFFF ^ (75422 >> 6) ^ ((75422 << 11) & 0xFFFFFFFF)
    assert a["samples"][0]["jitter"][0] == float(np.float32(np.float32(oracle.hash_u32(mix)) * np.float32(1.0 / 4294967296.0)))
    assert a["samples"][0]["jitter"][0] == oracle.rand2(3232319744.0, 0.37)


def test_path_traced_pixel_worked_in_float64(oracle):
    """computeLighting_optimized (compute.wgsl:344-410) with GI_SAMPLES = 2, GI_BOUNCES = 1, summed by hand in the same script:
    both bounce rays of pixel (4, 3) land on the large occluder's underside (t = 2.1143, 2.0881), whose direct terms are
    unshadowed (its own shadow rays leave it downwards): dS = 0.31935 and 0.18188, so
        indirect = (1, 0.84, 0) * (0.31935 + 0.18188) / 2 = (0.25061, 0.21052, 0);
    the primary hit's direct term on the 2 x 1 grid: samples 0..2 visible, 3 not (cos 0.016), only sample 1 hits the occluder:
        occ = 1, direct = 0.43269 * 0.75 = 0.32452;   col = albedo * direct + indirect = (0.57513, 0.48311, 0) -> (198, 183, 0, 255)."""
    d = _derivation()
    b = d.case_b()
    assert [r["hit_tri"] for r in b["bounces"]] == [0, 0] and [r["occ"] for r in b["bounces"]] == [0, 0]
    assert abs(b["bounces"][0]["direct"] - 0.31935) < 1e-5 and abs(b["bounces"][1]["direct"] - 0.18188) < 1e-5
    assert b["primary_occ"] == 1 and [(s["visible"], s["hits_occluder"]) for s in b["primary_samples"]] == \
        [(True, False), (True, True), (True, False), (False, False)]
    assert abs(b["col"][0] - (0.43269354682593947 * 0.75 + (0.3193483303026445 + 0.1818793674219388) / 2)) < 1e-12
    assert b["rgba8"] == [198, 183, 0, 255]
    nodes, tp, tn = d.flat_buffers()
    out, st = oracle.render(nodes, tp, tn, d.uniforms(d.LIGHT_B), mode=oracle.MODE_PT, gi_samples=2, gi_bounces=1, soft_samples=4,
                            rows=(3, 4, 1))
    x, y = b["pixel"]
    assert out["hit_tri"][y, x] == 1
    assert list(out["rgba8"][y, x]) == b["rgba8"]
    assert st["rays_bounce"] > 0 and st["rays_shadow"] > 0
