"""CPU-only: host-side product code (OBJ/MTL ingest, normalizeMesh, uniforms) against the oracle
restatement and hand-derived expectations."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_uniforms_match_oracle_bit_for_bit(rt, oracle):
    cases = [dict(), dict(cam_pos=(0, 0, 1.3)),
             dict(cam_pos=(0.5, 0.2, 2.0), yaw=0.3, pitch=-0.2, rot_deg=(10, 20, 30), scale=(1.5, 0.7, 2.0), translate=(0.1, -0.3, 0.5)),
             dict(yaw=3.0, pitch=1.5, rot_deg=(-180, 180, 90), scale=(0.1, 5, 1), time=7.0)]
    for kw in cases:
        for (w, h) in ((1920, 1080), (3840, 2160), (65, 65)):
            U = rt.make_uniforms(w, h, **kw)
            sl = tuple(kw.get("scale", (1, 1, 1))) + tuple(kw.get("rot_deg", (0, 0, 0))) + tuple(kw.get("translate", (0, 0, 0)))
            V = oracle.make_uniforms(cam_pos=kw.get("cam_pos", (0, 0, 5)), yaw=kw.get("yaw", 0), pitch=kw.get("pitch", 0), sliders=sl,
                                     width=w, height=h, time=kw.get("time", 0.0))
            assert np.array_equal(U.view(np.uint32), V.view(np.uint32))


def test_singular_model_matrix_gives_zero_inverse(rt):       # invert4 returns false, out stays zero (script.js:112,895)
    U = rt.make_uniforms(64, 64, scale=(0.0, 1.0, 1.0))
    assert np.all(U[20:36] == 0)


def test_plane_obj(rt, oracle, plane_obj_path):
    m = rt.load_obj(plane_obj_path)
    # 6 088 v / vn / vt, 6 298 triangular faces, 2 groups
    assert m.positions.shape == (6088, 3) and m.n_tri == 6298
    assert m.normals.shape == (6089, 3) and m.texcoords.shape == (6089, 2)     # + one dummy entry each
    assert m.vert_idx.max() == 6087 and m.norm_idx.max() <= 6087
    # first vertex line of the file: "v 26.080254 114.994545 40.519825"
    assert list(m.positions[0]) == [np.float32(26.080254), np.float32(114.994545), np.float32(40.519825)]
    # the float parser must agree with a correctly rounded parse on this file
    ref = []
    for line in open(plane_obj_path):
        if line.startswith("v "):
            ref.append([np.float32(float(x)) for x in line.split()[1:4]])
    assert np.array_equal(np.array(ref, np.float32), m.positions)
    raw = m.positions.copy()
    rt.normalize_mesh(m)
    assert np.array_equal(m.positions.view(np.uint32), oracle.normalize_mesh(raw).view(np.uint32))
    ext = m.positions.max(0) - m.positions.min(0)
    assert ext.max() == 1.0 and np.allclose(m.positions.max(0), -m.positions.min(0), atol=1e-6)


def test_2dplane_obj_and_mtl(rt, tmp_path):
    import shutil
    shutil.copy(os.path.join(GOLDEN, "2Dplane.obj"), tmp_path / "2Dplane.obj")
    (tmp_path / "2Dplane_material.mtl").write_text(
        "newmtl TexturedPlaneMaterial\nKd 0.5 0.25 1\nKe 0 0 0\nd 0.75\nillum 2\nmap_Kd texture.png\nmap_Bump -bm 1.0 n.png\n")
    m = rt.load_obj(tmp_path / "2Dplane.obj")
    assert m.n_tri == 4
    assert list(m.vert_idx) == [0, 1, 2, 1, 3, 2, 2, 1, 0, 2, 3, 1]
    assert list(m.norm_idx) == [0] * 6 + [1] * 6
    assert list(m.material_ids) == [0, 0, 0, 0]
    mat = m.materials[0]
    assert mat["name"] == "TexturedPlaneMaterial" and mat["map_Kd"] == "texture.png"
    assert mat["diffuse"] == (0.5, 0.25, 1.0) and mat["dissolve"] == 0.75 and mat["illum"] == 2


def test_obj_grammar(rt):
    txt = """# comment
v 0 0 0
v 1 0 0
v 1 1 0
v 0 2 0
v -.5 +3.e0 1E-1
vn 0 0 1
vt 0.5 0.5
f 1 2 3
f 1/1 2/1 3/1
f 1//1 2//1 3//1
f 1/1/1 2/1/1 3/1/1
f -5 -4 -3
f 1 2 3 4
f 1 2 3 4 5
f 1 2
"""
    m = rt.parse_obj(txt)
    # 5 triangles + quad (2) + pentagon fan (3); the 2-corner face is dropped
    assert m.n_tri == 10
    assert list(m.positions[4]) == [np.float32(-0.5), np.float32(3.0), np.float32(0.1)]
    assert list(m.vert_idx[:15]) == [0, 1, 2] * 5                      # incl. the negative (relative) indices
    assert list(m.norm_idx[0:3]) == [1, 1, 1]                          # no normal -> dummy entry (index n_nor)
    assert list(m.norm_idx[6:9]) == [0, 0, 0]
    assert list(m.tex_idx[0:3]) == [1, 1, 1] and list(m.tex_idx[3:6]) == [0, 0, 0]
    # quad 1-2-3-4: |v3-v1|^2 = 2, |v4-v2|^2 = 5 => split along 0-2 (tiny_obj_loader.h:1781-1791)
    assert list(m.vert_idx[15:21]) == [0, 1, 2, 0, 2, 3]
    assert list(m.vert_idx[21:30]) == [0, 1, 2, 0, 2, 3, 0, 3, 4]


def test_quad_split_other_diagonal(rt):
    m = rt.parse_obj("v 0 0 0\nv 2 0 0\nv 1 0.1 0\nv 0 1 0\nf 1 2 3 4\n")
    # |v3-v1|^2 = 1.01 >= |v4-v2|^2 = 5 ? no: 1.01 < 5 -> [0,1,2],[0,2,3]
    assert list(m.vert_idx) == [0, 1, 2, 0, 2, 3]
    m = rt.parse_obj("v 0 0 0\nv 1 0 0\nv 3 3 0\nv 0 1 0\nf 1 2 3 4\n")
    # |v3-v1|^2 = 18 >= |v4-v2|^2 = 2 -> [0,1,3],[1,2,3]
    assert list(m.vert_idx) == [0, 1, 3, 1, 2, 3]


def test_obj_errors(rt):
    with pytest.raises(rt.RtError):
        rt.parse_obj("v 0 0 0\nf 1 2 3\n")          # index beyond the vertices seen so far
    with pytest.raises(rt.RtError):
        rt.parse_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")   # zero vertex index is invalid
    with pytest.raises(rt.RtError):
        rt.load_obj("/nonexistent/file.obj")


def test_number_parser_matches_the_reference_arithmetic(rt):
    """tiny_obj_loader's tryParseDouble is NOT correctly rounded; the product restates its arithmetic."""
    import math
    lut = [1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001]

    def ref(s):
        sign = -1.0 if s[0] == "-" else 1.0
        s = s.lstrip("+-")
        mant, exp = (s.lower().split("e") + ["0"])[:2]
        ip, _, fp = mant.partition(".")
        m = 0.0
        for c in ip:
            m = m * 10 + int(c)
        for k, c in enumerate(fp, start=1):
            m += int(c) * (lut[k] if k < 8 else math.pow(10.0, -k))
        e = int(exp)
        return np.float32(sign * (math.ldexp(m * math.pow(5.0, e), e) if e else m))
    toks = ["0.1", "-12.345678901", "1e-3", "7.25E+2", "123456789.123456789", "0.000000119", "-.5", "3.", "1.17549435e-38"]
    m = rt.parse_obj("".join(f"v {t} 0 0\n" for t in toks))
    for t, got in zip(toks, m.positions[:, 0]):
        assert got == ref(t), t
