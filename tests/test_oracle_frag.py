"""The two older fragment-pipeline shaders (web/shader.wgsl, root shader.wgsl) as restated by the oracle.
PARITY UNPINNED by the reference (no fixtures exist for them either); what is checked here are hand-derived known answers
of their RNGs, the pinned acos, and structural properties that follow from the shader text."""
import numpy as np
import pytest

PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
FRAG = {3: dict(gi_samples=32, gi_bounces=3, soft_samples=8, max_stack=16),      # web/shader.wgsl:22-24,273
        4: dict(gi_samples=16, gi_bounces=1, soft_samples=8, max_stack=16)}      # shader.wgsl:24-25,191


def _pcg(x):            # web/shader.wgsl:176-180 by hand, Python integers
    v = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    return ((v >> ((v >> 28) + 4)) ^ v) & 0xFFFFFFFF


def _xs(x):             # shader.wgsl:320-326
    v = x
    v = (((v >> 16) ^ v) * 0x45D9F3B) & 0xFFFFFFFF
    v = (((v >> 16) ^ v) * 0x45D9F3B) & 0xFFFFFFFF
    return ((v >> 16) ^ v) & 0xFFFFFFFF


def test_frag_rngs_known_answers(oracle):
    assert _pcg(0) == ((2891336453 >> ((2891336453 >> 28) + 4)) ^ 2891336453)
    assert _xs(0) == 0
    for x in (0, 1, 2, 0x3F800000, 0xDEADBEEF, 0xFFFFFFFF):
        assert oracle.frag_hash(1, x) == _pcg(x)
        assert oracle.frag_hash(2, x) == _xs(x)
    # rand2 hashes the seeds' BIT PATTERNS: xi ^ (yi << 16)   (web/shader.wgsl:181-185)
    for v, h in ((1, _pcg), (2, _xs)):
        for sx, sy in ((0.0, 17.0), (3.0, 18.0), (0.5, 0.25), (37.0, 2.0)):
            xi = int(np.float32(sx).view(np.uint32)); yi = int(np.float32(sy).view(np.uint32))
            want = np.float32(np.float32(h(xi ^ ((yi << 16) & 0xFFFFFFFF))) * np.float32(1.0 / 4294967296.0))
            assert oracle.frag_rand2(v, sx, sy) == float(want)


def test_pinned_acos_accuracy(oracle):
    xs = np.concatenate([np.linspace(-1, 1, 4001), [0.25, np.sqrt(0.1875), 0.5, -0.5]]).astype(np.float32)
    got = np.array([oracle.acos(float(x)) for x in xs])
    want = np.arccos(xs.astype(np.float64))
    assert np.max(np.abs(got - want)) < 4e-7 * np.pi
    assert oracle.acos(1.0) == 0.0 and oracle.acos(2.0) == 0.0          # clamped


@pytest.mark.parametrize("mode", [3, 4])
def test_frag_image_properties(oracle, rt, plane_mesh, mode):
    b = oracle.build_lbvh(plane_mesh.positions, plane_mesh.normals, plane_mesh.vert_idx, plane_mesh.norm_idx)
    w, h = 96, 64
    U = rt.make_uniforms(w, h, **PLANE_VIEW)
    out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, n_threads=8, **FRAG[mode])
    hit = out["hit_tri"] != 0xFFFFFFFF
    assert 0.2 < hit.mean() < 0.9
    # a miss is the sky (0.6, 0.8, 1, 1) as rgba8unorm  (web/shader.wgsl:391, shader.wgsl:408)
    assert np.all(out["rgba8"][~hit] == np.array([153, 204, 255, 255], np.uint8))
    # albedo (1, 0.84, 0): blue stays 0 on every hit, alpha 1
    assert np.all(out["rgba8"][hit][:, 2] == 0) and np.all(out["rgba8"][..., 3] == 255)
    assert st["rays_primary"] == w * h and st["pixels"] == w * h
    # the fragment pipeline's y runs the other way: row y sees what the compute shader's row h-1-y sees.  The triangle
    # tests differ only in thresholds (1e-8 / 1e-6 vs 1e-10), which no triangle of this mesh comes near.
    ref, _ = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=0, max_stack=16)
    same = out["hit_tri"] == ref["hit_tri"][::-1]
    assert same.mean() > 0.995
    # rows restriction renders exactly those rows
    part, _ = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, rows=(8, 24, 1), **FRAG[mode])
    assert np.array_equal(part["rgba8"][8:24], out["rgba8"][8:24])


def test_frag_web_random_numbers_depend_on_the_sample_only(oracle):
    """web/shader.wgsl:410-411 seeds its GI directions with (i, bounce) only — no pixel term — and rand2 (:181-185) keeps
    just the LOW 16 bits of the second seed's bit pattern (`yi << 16`), which are zero for every small whole number: the
    bounce index never reaches the hash.  So each sample i has ONE (rx, ry) pair for all of its bounces and all pixels."""
    for i in range(6):
        rx = {oracle.frag_rand2(1, float(i), float(b) + 17.0) for b in range(3)}
        ry = {oracle.frag_rand2(1, float(i) + 37.0, float(b)) for b in range(3)}
        assert len(rx) == 1 and len(ry) == 1
    assert len({oracle.frag_rand2(1, float(i), 17.0) for i in range(32)}) == 32


def test_frag_root_hemisphere_table_is_exact():
    """shader.wgsl:45-50 : the 16 stratified samples are ((idx % 8) * 2 + 1) / 16 and ((idx / 8) * 2 + 1) / 16."""
    table = [(0.0625, 0.0625), (0.1875, 0.0625), (0.3125, 0.0625), (0.4375, 0.0625), (0.5625, 0.0625), (0.6875, 0.0625),
             (0.8125, 0.0625), (0.9375, 0.0625), (0.0625, 0.1875), (0.1875, 0.1875), (0.3125, 0.1875), (0.4375, 0.1875),
             (0.5625, 0.1875), (0.6875, 0.1875), (0.8125, 0.1875), (0.9375, 0.1875)]
    for idx, (x, y) in enumerate(table):
        assert np.float32(((idx & 7) * 2 + 1) / 16.0) == np.float32(x) and np.float32(((idx >> 3) * 2 + 1) / 16.0) == np.float32(y)


def test_frag_rejects_more_taps_than_the_tables_hold(oracle, rt, plane_mesh):
    b = oracle.build_lbvh(plane_mesh.positions, plane_mesh.normals, plane_mesh.vert_idx, plane_mesh.norm_idx)
    U = rt.make_uniforms(16, 16, **PLANE_VIEW)
    with pytest.raises(RuntimeError):
        oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=3, soft_samples=9)
    with pytest.raises(RuntimeError):
        oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=4, gi_samples=17, soft_samples=8)
