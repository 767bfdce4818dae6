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
    # 5 triangles + quad (2) + pentagon (3 ears, here the same as a fan: see test_polygon_ear_clipping); the 2-corner face is dropped
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


def _tris(m):
    return [tuple(int(v) for v in m.vert_idx[3 * i:3 * i + 3]) for i in range(m.n_tri)]


def test_polygon_ear_clipping_convex_pentagon(rt):
    """tinyobj's built-in ear clipping (tiny_obj_loader.h:1945-2191), worked by hand.  Convex CCW pentagon in z = 0 that does
    NOT touch the origin: P0(1,1) P1(3,1) P2(4,3) P3(2,4) P4(0,3).
    axes: corner (P0,P1,P2): e0 = (2,0,0), e1 = (1,2,0) -> |cross| = (0, 0, 4): z largest -> work in (x, y).
    g = 0: ear (P0,P1,P2): cross = 2*2 - 0*1 = 4, "area" = (x0*y1 - y0*x1)/2 = (1*1 - 1*3)/2 = -1 -> cross*area < 0: skipped, g = 1.
    g = 1: ear (P1,P2,P3): e0 = (1,2), e1 = (-2,1): cross = 5, area = (3*3 - 1*4)/2 = 2.5; P4 and P0 lie outside -> emit (1,2,3),
           drop P2: ring P0 P1 P3 P4.
    g = 1: ear (P1,P3,P4): e0 = (-1,3), e1 = (-2,-1): cross = 7, area = (3*4 - 1*2)/2 = 5; P0 outside -> emit (1,3,4), drop P3.
    rest: (0,1,4).  (A fan from P0 - what this parser did before - gives (0,1,2),(0,2,3),(0,3,4).)"""
    m = rt.parse_obj("v 1 1 0\nv 3 1 0\nv 4 3 0\nv 2 4 0\nv 0 3 0\nf 1 2 3 4 5\n")
    assert _tris(m) == [(1, 2, 3), (1, 3, 4), (0, 1, 4)]


def test_polygon_ear_clipping_concave_hexagon(rt):
    """L-shaped hexagon in the plane x = 2, (y, z) = Q0(0,0) Q1(4,0) Q2(4,4) Q3(3,4) Q4(3,1) Q5(0,1); Q4 is the reflex corner.
    axes: corner (Q0,Q1,Q2): e0 = (0,4,0), e1 = (0,0,4) -> |cross| = (16, 0, 0): x largest -> work in (y, z).
    g = 0: (Q0,Q1,Q2): cross = 16, area = 0 -> candidate; Q4(3,1) is inside the triangle (0,0),(4,0),(4,4) -> overlap, g = 1.
    g = 1: (Q1,Q2,Q3) = (4,0),(4,4),(3,4): cross = 0*0 - 4*(-1) = 4, area = (4*4 - 0*4)/2 = 8; Q4, Q5, Q0 outside -> emit (1,2,3);
           ring Q0 Q1 Q3 Q4 Q5.
    g = 1: (Q1,Q3,Q4) = (4,0),(3,4),(3,1): cross = (-1)(-3) - 4*0 = 3, area = (4*4 - 0*3)/2 = 8; Q5, Q0 outside -> emit (1,3,4);
           ring Q0 Q1 Q4 Q5.
    g = 1: (Q1,Q4,Q5) = (4,0),(3,1),(0,1): cross = (-1)*0 - 1*(-3) = 3, area = (4*1 - 0*3)/2 = 2; Q0 outside -> emit (1,4,5).
    rest: (0,1,5).  Areas 2 + 1.5 + 1.5 + 2 = 7 = the L's area (4x1 + 1x3): nothing outside the polygon is covered,
    which a fan from Q0 would do."""
    m = rt.parse_obj("v 2 0 0\nv 2 4 0\nv 2 4 4\nv 2 3 4\nv 2 3 1\nv 2 0 1\nf 1 2 3 4 5 6\n")
    assert _tris(m) == [(1, 2, 3), (1, 3, 4), (1, 4, 5), (0, 1, 5)]
    P = m.positions.astype(np.float64)
    area = sum(0.5 * np.linalg.norm(np.cross(P[b] - P[a], P[c] - P[a])) for a, b, c in _tris(m))
    assert area == 7.0


def test_polygon_ear_clipping_collinear_points(rt):
    """Pentagon in the plane y = 0 whose first three corners are collinear: (x, z) = R0(1,1) R1(2,1) R2(3,1) R3(3,3) R4(1,3).
    axes: corner (R0,R1,R2) has a zero cross product and is passed over; corner (R1,R2,R3): e0 = (1,0,0), e1 = (0,0,2) ->
          |cross| = (0, 2, 0): neither x nor z is largest -> work in (x, z).
    g = 0: (R0,R1,R2): cross = 0 -> cross*area = 0 is not < 0, no other corner inside the (empty) triangle -> tinyobj emits the
           DEGENERATE triangle (0,1,2) and drops R1; ring R0 R2 R3 R4.
    g = 0: (R0,R2,R3) = (1,1),(3,1),(3,3): cross = 4, area = (1*1 - 1*3)/2 = -1 -> skipped, g = 1.
    g = 1: (R2,R3,R4) = (3,1),(3,3),(1,3): cross = 0*0 - 2*(-2) = 4, area = (3*3 - 1*3)/2 = 3; R0 outside -> emit (2,3,4).
    rest: (0,2,4)."""
    m = rt.parse_obj("v 1 0 1\nv 2 0 1\nv 3 0 1\nv 3 0 3\nv 1 0 3\nf 1 2 3 4 5\n")
    assert _tris(m) == [(0, 1, 2), (2, 3, 4), (0, 2, 4)]
    # normals / texcoords / materials follow the corners
    m = rt.parse_obj("v 1 0 1\nv 2 0 1\nv 3 0 1\nv 3 0 3\nv 1 0 3\nvn 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 1\n"
                     "usemtl a\nf 1/1/1 2/2/2 3/1/1 4/2/2 5/1/1\n", "newmtl a\nKd 1 0 0\n")
    assert list(m.norm_idx) == [0, 1, 0, 0, 1, 0, 0, 0, 0] and list(m.tex_idx) == [0, 1, 0, 0, 1, 0, 0, 0, 0]
    assert list(m.material_ids) == [0, 0, 0]


def test_mtl_unknown_parameters_are_kept(rt):
    """material_t::unknown_parameter (tiny_obj_loader.h:2749-2762): a line whose key tinyobj has no field for is stored as
    key -> rest of the line; the key ends at the first ' ' (at the first tab only when the line holds no ' '), the value
    is everything behind that ONE character (leading blanks stay, trailing blanks of the line are trimmed before), the
    first value of a repeated key wins, keys come back in std::map order; keys tinyobj parses itself (Ka, Ns, Ni, Pr, ...)
    are not unknown even where this library keeps no field for them."""
    mtl = ("newmtl a\nKd 1 0 0\nKa 0.1 0.1 0.1\nNs 10\nNi 1.5\nPr 0.5\nzeta  two  words  \nalpha 1\nalpha 2\n"
           "tab\tsep\nmixed\tkey val\nnovalue\n# comment\n"
           "newmtl b\nmap_Kd t.png\nfoo bar\n")
    m = rt.parse_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n", mtl)
    a, b = m.materials
    assert a["unknown_parameter"] == {"alpha": "1", "mixed\tkey": "val", "tab": "sep", "zeta": " two  words"}
    assert list(a["unknown_parameter"]) == ["alpha", "mixed\tkey", "tab", "zeta"]
    assert b["unknown_parameter"] == {"foo": "bar"} and b["map_Kd"] == "t.png"
    assert a["diffuse"] == (1.0, 0.0, 0.0)


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


def test_load_texture_and_bind_obj_surface(rt, tmp_path):
    """The file side of RT_FLAG_TEXTURED: binary PPM without PIL, PNG / JPEG through it; bind_obj_surface gives every MTL material
    its Kd, a texture slot where map_Kd names a file that exists (shared between materials that name the same file), -1 else."""
    img = np.random.default_rng(1).integers(0, 256, (5, 7, 3)).astype(np.uint8)
    (tmp_path / "a.ppm").write_bytes(b"P6\n# a comment\n7 5\n255\n" + img.tobytes())
    assert np.array_equal(rt.load_texture(str(tmp_path / "a.ppm")), img)
    with pytest.raises(ValueError):
        (tmp_path / "short.ppm").write_bytes(b"P6\n7 5\n255\n" + img.tobytes()[:50])
        rt.load_texture(str(tmp_path / "short.ppm"))
    try:
        from PIL import Image
        Image.fromarray(img).save(str(tmp_path / "b.png"))
        assert np.array_equal(rt.load_texture(str(tmp_path / "b.png")), img)
    except ImportError:
        pass
    obj = "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl one\nf 1/1 2/2 3/3\nusemtl two\nf 2/1 4/2 3/3\nusemtl three\nf 1/1 4/2 3/3\n"
    mtl = "newmtl one\nKd 1 0.5 0.25\nmap_Kd a.ppm\nnewmtl two\nKd 0.1 0.2 0.3\nmap_Kd missing.jpg\nnewmtl three\nKd 0.9 0.9 0.9\nmap_Kd a.ppm\n"
    mesh = rt.parse_obj(obj, mtl)
    assert [m["map_Kd"] for m in mesh.materials] == ["a.ppm", "missing.jpg", "a.ppm"]

    class Ctx:
        def __init__(self): self.calls = []
        def set_surface(self, m): self.calls.append(("surface",))
        def set_texture(self, slot, a): self.calls.append(("texture", slot, a.shape))
        def set_materials(self, recs): self.calls.append(("materials", [(tuple(round(float(x), 3) for x in kd), s) for kd, s in recs]))
    c = Ctx()
    bound = rt.bind_obj_surface(c, mesh, str(tmp_path))
    assert bound == [("a.ppm", 0)]
    assert c.calls == [("surface",), ("texture", 0, (5, 7, 3)),
                       ("materials", [((1.0, 0.5, 0.25), 0), ((0.1, 0.2, 0.3), -1), ((0.9, 0.9, 0.9), 0)])]
