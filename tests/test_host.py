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


# ---- LoadMtl restated branch by branch: every expected value below is worked from the TEXT of tiny_obj_loader.h (no code of it
# ---- was run: SURVEY §8c), the line numbers are those of /root/reference/tiny_obj_loader.h
TRI = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"


def _mats(rt, mtl, obj=TRI + "f 1 2 3\n"):
    return rt.parse_obj(obj, mtl).materials


def test_mtl_default_diffuse_is_zero_without_kd_and_map_kd(rt):
    """InitMaterial sets diffuse to 0 (:1588); nothing else touches it when the material has neither Kd nor map_Kd."""
    (m,) = _mats(rt, "newmtl bare\nKa 0.25 0.5 0.75\nNs 8\n")
    assert m["diffuse"] == (0.0, 0.0, 0.0) and m["ambient"] == (0.25, 0.5, 0.75) and m["shininess"] == 8.0
    assert m["dissolve"] == 1.0 and m["ior"] == 1.0 and m["illum"] == 0 and m["map_Kd"] == ""      # :1593-1596


def test_mtl_map_kd_sets_0_6_only_when_no_kd_came_before(rt):
    """:2609-2625: `if (!has_kd) diffuse = 0.6` runs when the map_Kd LINE is parsed: a Kd line in front of it survives, a Kd
    line behind it overwrites the 0.6 (:2432-2442)."""
    a, b, c = _mats(rt, "newmtl only_map\nmap_Kd t.png\n"
                        "newmtl map_then_kd\nmap_Kd t.png\nKd 0.125 0.25 0.5\n"
                        "newmtl kd_then_map\nKd 0.125 0.25 0.5\nmap_Kd t.png\n")
    assert a["diffuse"] == tuple(float(np.float32(0.6)) for _ in range(3))
    assert b["diffuse"] == (0.125, 0.25, 0.5) and c["diffuse"] == (0.125, 0.25, 0.5)
    assert a["map_Kd"] == b["map_Kd"] == c["map_Kd"] == "t.png"


def test_mtl_texture_options_are_stripped_from_the_name(rt):
    """ParseTextureNameAndOption (:1411-1528): options in front of the name are consumed into texture_option_t; the name is
    the rest of the line from the first non-option token, blanks included."""
    (m,) = _mats(rt, "newmtl m\nmap_Kd -s 2 2 1 -o 0.5 0 0 file name with spaces.png\n"
                     "map_Bump -bm 0.25 -imfchan r -clamp on n.png\n"
                     "refl -type cube_left -texres 256 -mm 0.5 2 -blendu off -blendv bogus -boost 3 -t 1 2 3 -colorspace sRGB r.hdr -s 9 9 9\n")
    assert m["map_Kd"] == "file name with spaces.png"
    o = m["map_Kd_opt"]
    assert o["scale"] == (2.0, 2.0, 1.0) and o["origin_offset"] == (0.5, 0.0, 0.0) and o["imfchan"] == "m" and o["clamp"] == 0
    assert m["map_Bump"] == "n.png"
    b = m["map_Bump_opt"]
    assert b["bump_multiplier"] == 0.25 and b["imfchan"] == "r" and b["clamp"] == 1 and b["scale"] == (1.0, 1.0, 1.0)
    # options BEHIND the name are part of the name (:1509-1512 reads to the end of the line)
    assert m["refl"] == "r.hdr -s 9 9 9"
    r = m["refl_opt"]
    assert r["type"] == 6 and r["texture_resolution"] == 256 and (r["brightness"], r["contrast"]) == (0.5, 2.0)
    assert r["blendu"] == 0 and r["blendv"] == 1                         # parseOnOff: neither "on" nor "off" -> its default, true
    assert r["sharpness"] == 3.0 and r["turbulence"] == (1.0, 2.0, 3.0) and r["colorspace"] == "sRGB" and r["scale"] == (1.0, 1.0, 1.0)
    # defaults of InitTexOpt (:1530-1558) where a map has no line at all; the bump slot alone starts with imfchan 'l'
    d = m["map_Ks_opt"]
    assert m["map_Ks"] == "" and d["imfchan"] == "m" and d["texture_resolution"] == -1 and d["blendu"] == 1 and d["scale"] == (1.0, 1.0, 1.0)
    assert _mats(rt, "newmtl m\nKd 1 1 1\n")[0]["map_Bump_opt"]["imfchan"] == "l"


def test_mtl_options_without_a_name_leave_the_name_alone(rt):
    """No filename token -> ParseTextureNameAndOption returns false and *texname keeps its value (:1519-1527); map_d first
    stores the raw rest of the line (:2669), so there the options themselves stay as the name."""
    (m,) = _mats(rt, "newmtl m\nmap_Kd first.png\nmap_Kd -clamp on\nmap_d -clamp on\n")
    assert m["map_Kd"] == "first.png" and m["map_Kd_opt"]["clamp"] == 1
    assert m["map_d"] == "-clamp on" and m["map_d_opt"]["clamp"] == 1
    # a name that merely starts like an option is a name: IS_SPACE(token[2]) fails for "-old.png" (:1448)
    assert _mats(rt, "newmtl m\nmap_Kd -old.png\n")[0]["map_Kd"] == "-old.png"


def test_mtl_d_wins_against_tr(rt):
    """:2506-2541: `Tr` sets dissolve = 1 - Tr unless a `d` line came before it in this material; a later `d` always overwrites."""
    a, b, c = _mats(rt, "newmtl d_then_tr\nd 0.75\nTr 0.5\n" "newmtl tr_then_d\nTr 0.5\nd 0.75\n" "newmtl tr_only\nTr 0.25\n")
    assert a["dissolve"] == 0.75 and b["dissolve"] == 0.75 and c["dissolve"] == 0.75
    # parseReal's default is 0.0 for `d` too (:1124, :2509): an unparsable value is 0, not 1
    assert _mats(rt, "newmtl m\nd x\n")[0]["dissolve"] == 0.0


def test_usemtl_unknown_name_is_minus_one_and_names_are_one_word(rt):
    """usemtl (:3200-3233): parseString = ONE word, looked up in material_map; not found -> material id -1 (and a warning
    there).  newmtl (:2403-2414) takes one word as well; std::map::insert keeps the FIRST material of a repeated name."""
    mtl = "newmtl red paint\nKd 1 0 0\nnewmtl blue\nKd 0 0 1\nnewmtl red\nKd 0.5 0 0\n"
    obj = TRI + "usemtl red\nf 1 2 3\nusemtl nope\nf 1 2 3\nusemtl blue extra words\nf 1 2 3\nusemtlred\nf 1 2 3\nusemtl\nf 1 2 3\n"
    m = rt.parse_obj(obj, mtl)
    assert [x["name"] for x in m.materials] == ["red", "blue", "red"]
    assert list(m.material_ids) == [0, -1, 1, 0, -1]            # "usemtlred": no blank is asked for behind the key (:3200)


def test_mtl_without_newmtl_yields_the_default_material(rt):
    """LoadMtl creates a default material anyway (:2328) and always flushes the last one, named or not (:2764-2767); lines in
    front of the first newmtl land in that nameless material, which a first newmtl then discards (:2388)."""
    (only,) = _mats(rt, "Kd 0.25 0.5 0.75\n")
    assert only["name"] == "" and only["diffuse"] == (0.25, 0.5, 0.75)
    (named,) = _mats(rt, "Kd 0.25 0.5 0.75\nnewmtl a\nKs 1 1 1\n")
    assert named["name"] == "a" and named["diffuse"] == (0.0, 0.0, 0.0) and named["specular"] == (1.0, 1.0, 1.0)


def test_mtl_pbr_and_remaining_fields(rt):
    (m,) = _mats(rt, "newmtl m\nKs 0.1 0.2 0.3\nKt 0.4 0.5 0.6\nNi 1.45\nNs 96\nPr 0.5\nPm 0.25\nPs 0.125\nPc 0.75\nPcr 0.0625\n"
                     "aniso 0.5\nanisor 0.25\nmap_Pr r.png\nmap_Pm m.png\nmap_Ps s.png\nmap_Ke e.png\nnorm n.png\nmap_Ka a.png\n"
                     "map_Ns ns.png\ndisp d.png\nmap_Disp D.png\nTf 1 1 1\n")
    f32 = lambda x: float(np.float32(x))
    assert m["specular"] == (f32(0.1), f32(0.2), f32(0.3)) and m["transmittance"] == (1.0, 1.0, 1.0)       # Tf after Kt: same field (:2457)
    assert m["ior"] == f32(1.45) and m["shininess"] == 96.0
    assert (m["roughness"], m["metallic"], m["sheen"], m["clearcoat_thickness"], m["clearcoat_roughness"]) == (0.5, 0.25, 0.125, 0.75, 0.0625)
    assert (m["anisotropy"], m["anisotropy_rotation"]) == (0.5, 0.25)
    assert [m[k] for k in ("map_Pr", "map_Pm", "map_Ps", "map_Ke", "norm", "map_Ka", "map_Ns", "disp")] == \
        ["r.png", "m.png", "s.png", "e.png", "n.png", "a.png", "ns.png", "D.png"]
    assert m["unknown_parameter"] == {}


def test_mtllib_takes_the_first_file_that_opens(rt, tmp_path):
    """:3236-3303: the names on a mtllib line are split at ' ' ('\\' escapes), tried in order, the first that opens ends the search."""
    (tmp_path / "b.mtl").write_text("newmtl fromb\nKd 0 1 0\n")
    (tmp_path / "c.mtl").write_text("newmtl fromc\nKd 0 0 1\n")
    (tmp_path / "with space.mtl").write_text("newmtl spaced\nKd 1 1 0\n")
    (tmp_path / "o.obj").write_text("mtllib missing.mtl b.mtl c.mtl\nmtllib with\\ space.mtl\nmtllib b.mtl\n" + TRI + "usemtl fromb\nf 1 2 3\nusemtl fromc\nf 1 2 3\nusemtl spaced\nf 1 2 3\n")
    m = rt.load_obj(tmp_path / "o.obj")
    assert [x["name"] for x in m.materials] == ["fromb", "spaced"] and list(m.material_ids) == [0, -1, 1]


def test_reference_mtl_files_parse_to_the_values_written_out_by_hand(rt):
    """The reference's three MTL files (models/*.mtl, data fixtures copied by tests/golden/make_textures.py), expected values
    read off their text with the rules above."""
    def load(name):
        return _mats(rt, open(os.path.join(GOLDEN, name)).read())
    one = (1.0, 1.0, 1.0)
    p85, p86 = load("plane_material.mtl")
    for m, kd in ((p85, "Material_85_baseColor.jpg"), (p86, "Material_86_baseColor.png")):
        # `Kd 1 1 1   # fallback ...`: parseReal3 reads three tokens, the rest of the line is never looked at (:2434-2441)
        assert m["diffuse"] == one and m["emission"] == (0.0, 0.0, 0.0) and m["illum"] == 2
        assert m["dissolve"] == 1.0                              # `d 1` then `Tr 0`: d wins
        assert m["map_Kd"] == kd and m["map_Bump"] == "Material_85_normal.jpg"
        # map_Pr / map_Pm ARE fields of material_t in this tinyobj version (:2705-2719) - the MTL's own comment, which expects
        # them in unknown_parameter, describes an older one
        assert m["map_Pr"] == m["map_Pm"] == "Material_85_metallicRoughness_metal.jpg" and m["unknown_parameter"] == {}
    assert (p85["name"], p86["name"]) == ("Material_85", "Material_86")
    d1, d2 = load("dragon_material.mtl")
    assert d1["name"] == "MI_M_B_44_Qishilong_body02_Inst" and d2["name"] == "MI_M_B_44_Qishilong_body02_2_Inst"
    assert d1["map_Kd"] == "MI_M_B_44_Qishilong_body02_Inst_diffuse.jpg" and d1["norm"] == "MI_M_B_44_Qishilong_body02_Inst_normal.jpg"
    assert d2["map_Kd"] == "MI_M_B_44_Qishilong_body02_2_Inst_diffuse.jpg" and d2["norm"] == ""
    for m in (d1, d2):
        assert m["diffuse"] == one and m["dissolve"] == 1.0 and m["illum"] == 2 and (m["roughness"], m["metallic"]) == (0.0, 0.0)
    (q,) = load("2Dplane_material.mtl")
    assert q["name"] == "TexturedPlaneMaterial" and q["map_Kd"] == "texture.png" and q["diffuse"] == one and q["dissolve"] == 1.0
