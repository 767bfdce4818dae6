"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package raytracer_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MODE_PRIMARY, MODE_SHADOW, MODE_PT = 0, 1, 2
MODE_FRAG_WEB, MODE_FRAG_ROOT = 3, 4      # web/shader.wgsl, shader.wgsl


class RenderOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mode", "gi_samples", "gi_bounces", "soft_samples", "max_stack",
        "row_begin", "row_end", "row_step", "n_threads", "node_encoding")]


class RenderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests",
        "hit_updates", "pixels", "stack_drops")] + [("seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class _Surface(C.Structure):
    _fields_ = [("tri_uv", C.c_void_p), ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("texels", C.c_void_p),
                ("tex_table", C.c_void_p), ("n_textures", C.c_uint32)]


class Surface:
    """Material/texture extension inputs in the oracle's layout (rt_oracle.h:orc_surface).
    tri_uv [T,8] BVH order; materials: list of (kd_rgb, texture_slot or -1); textures: list of uint8 [H,W,3] arrays."""

    def __init__(self, tri_uv, materials, textures):
        self.tri_uv = np.ascontiguousarray(tri_uv, np.float32).reshape(-1, 8)
        m = np.zeros((max(len(materials), 1), 4), np.float32)
        for i, (kd, slot) in enumerate(materials):
            m[i, :3] = kd
            m[i, 3:4].view(np.uint32)[0] = 0xFFFFFFFF if slot is None or slot < 0 else slot
        self.materials, self.n_materials = m, len(materials)
        table, tex = [], []
        off = 0
        for t in textures:
            t = np.ascontiguousarray(t, np.uint8)
            h, w = t.shape[:2]
            px = t[..., 0].astype(np.uint32) | (t[..., 1].astype(np.uint32) << 8) | (t[..., 2].astype(np.uint32) << 16) | np.uint32(0xFF000000)
            tex.append(px.reshape(-1))
            table.append((off, w, h, 0))
            off += w * h
        self.texels = np.concatenate(tex) if tex else np.zeros(1, np.uint32)
        self.table = np.asarray(table if table else [(0, 0, 0, 0)], np.uint32)
        self.n_textures = len(textures)
        self.c = _Surface(self.tri_uv.ctypes.data, self.materials.ctypes.data, self.n_materials, self.texels.ctypes.data,
                          self.table.ctypes.data, self.n_textures)


def gather_tri_uv(texcoords, tex_idx, material_ids, order):
    tc = np.ascontiguousarray(texcoords, np.float32).reshape(-1)
    ti = np.ascontiguousarray(tex_idx, np.uint32).reshape(-1)
    o = np.ascontiguousarray(order, np.uint32)
    mid = None if material_ids is None else np.ascontiguousarray(material_ids, np.int32)
    out = np.zeros((o.size, 8), np.float32)
    lib().orc_gather_tri_uv(tc.ctypes.data_as(C.c_void_p), ti.ctypes.data_as(C.c_void_p),
                            None if mid is None else mid.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), o.size,
                            out.ctypes.data_as(C.c_void_p))
    return out


def build(force=False):
    """Compile liboracle.so with the committed recipe (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(_LIB_PATH)
        f32p, u32p, i32p = (np.ctypeslib.ndpointer(d, flags="C_CONTIGUOUS") for d in (np.float32, np.uint32, np.int32))
        L.orc_expand_bits.restype = C.c_uint32; L.orc_expand_bits.argtypes = [C.c_uint32]
        L.orc_quantize.restype = C.c_uint32; L.orc_quantize.argtypes = [C.c_float]
        L.orc_hash_u32.restype = C.c_uint32; L.orc_hash_u32.argtypes = [C.c_uint32]
        L.orc_rand2.restype = C.c_float; L.orc_rand2.argtypes = [C.c_float, C.c_float]
        L.orc_sincos.restype = None; L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_pow.restype = C.c_float; L.orc_pow.argtypes = [C.c_float, C.c_float]
        L.orc_acos.restype = C.c_float; L.orc_acos.argtypes = [C.c_float]
        L.orc_frag_hash.restype = C.c_uint32; L.orc_frag_hash.argtypes = [C.c_int, C.c_uint32]
        L.orc_frag_rand2.restype = C.c_float; L.orc_frag_rand2.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_normalize_mesh.restype = None; L.orc_normalize_mesh.argtypes = [f32p, C.c_uint32]
        L.orc_centroid_bounds.restype = None; L.orc_centroid_bounds.argtypes = [f32p, u32p, C.c_uint32, f32p, f32p]
        L.orc_morton.restype = None; L.orc_morton.argtypes = [f32p, u32p, C.c_uint32, f32p, f32p, u32p]
        L.orc_sort.restype = None; L.orc_sort.argtypes = [u32p, C.c_uint32, u32p, u32p]
        L.orc_karras.restype = None; L.orc_karras.argtypes = [u32p, C.c_uint32, i32p, i32p, i32p]
        L.orc_build_lbvh.restype = None
        L.orc_build_lbvh.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, u32p, u32p, C.c_uint32,
                                     f32p, f32p, f32p, u32p, u32p, C.POINTER(C.c_double)]
        L.orc_build_lbvh_enc.restype = None
        L.orc_build_lbvh_enc.argtypes = L.orc_build_lbvh.argtypes + [C.c_int]
        L.orc_build_sah.restype = C.c_int
        L.orc_build_sah.argtypes = [f32p, f32p, u32p, u32p, C.c_uint32, C.c_int, f32p, C.c_uint32, C.POINTER(C.c_uint32),
                                    f32p, f32p, u32p, C.POINTER(C.c_double)]
        L.orc_set_surface.restype = None; L.orc_set_surface.argtypes = [C.c_void_p]
        L.orc_gather_tri_uv.restype = None
        L.orc_gather_tri_uv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_make_uniforms.restype = None
        L.orc_make_uniforms.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double),
                                        C.c_uint32, C.c_uint32, C.c_float, f32p]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [f32p, C.c_uint32, f32p, f32p, C.c_uint32, f32p, C.POINTER(RenderOpts),
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RenderStats)]
        _lib = L
    return _lib


def expand_bits(v): return int(lib().orc_expand_bits(v))
def quantize(x): return int(lib().orc_quantize(x))
def hash_u32(x): return int(lib().orc_hash_u32(x))
def rand2(x, y): return float(lib().orc_rand2(x, y))
def pow_(x, y): return float(lib().orc_pow(x, y))
def acos(x): return float(lib().orc_acos(x))
def frag_hash(variant, x): return int(lib().orc_frag_hash(variant, x & 0xFFFFFFFF))
def frag_rand2(variant, sx, sy): return float(lib().orc_frag_rand2(variant, sx, sy))


def sincos(theta):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(theta, C.byref(s), C.byref(c))
    return s.value, c.value


def normalize_mesh(pos):
    p = np.ascontiguousarray(pos, np.float32).reshape(-1).copy()
    lib().orc_normalize_mesh(p, p.size // 3)
    return p.reshape(-1, 3)


def _mesh(pos, vidx):
    return (np.ascontiguousarray(pos, np.float32).reshape(-1), np.ascontiguousarray(vidx, np.uint32).reshape(-1))


def centroid_bounds(pos, vidx):
    p, v = _mesh(pos, vidx)
    bmin, inv = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_centroid_bounds(p, v, v.size // 3, bmin, inv)
    return bmin, inv


def morton(pos, vidx, bmin, inv):
    p, v = _mesh(pos, vidx)
    codes = np.zeros(v.size // 3, np.uint32)
    lib().orc_morton(p, v, v.size // 3, np.ascontiguousarray(bmin, np.float32), np.ascontiguousarray(inv, np.float32), codes)
    return codes


def sort(codes):
    c = np.ascontiguousarray(codes, np.uint32)
    s, o = np.zeros_like(c), np.zeros_like(c)
    lib().orc_sort(c, c.size, s, o)
    return s, o


def karras(sorted_codes):
    k = np.ascontiguousarray(sorted_codes, np.uint32)
    T = k.size
    c0, c1 = np.zeros(max(T - 1, 1), np.int32), np.zeros(max(T - 1, 1), np.int32)
    par = np.full(max(2 * T - 1, 1), -1, np.int32)
    lib().orc_karras(k, T, c0, c1, par)
    return c0[:max(T - 1, 0)], c1[:max(T - 1, 0)], par


def build_lbvh(pos, nor, vidx, nidx, encoding=1):
    """Returns dict(nodes[N,8], tri_pos[T,3,4], tri_nor[T,3,4], codes[T], order[T], seconds).
    encoding: 1 = v1 node records (reference), 2 = v2 (u32 bit-casts; needed above 2^24 nodes)."""
    p, v = _mesh(pos, vidx)
    n, ni = _mesh(nor, nidx)
    T = v.size // 3
    nodes = np.zeros((2 * T - 1) * 8, np.float32)
    tp, tn = np.zeros(T * 12, np.float32), np.zeros(T * 12, np.float32)
    codes, order = np.zeros(T, np.uint32), np.zeros(T, np.uint32)
    sec = C.c_double()
    lib().orc_build_lbvh_enc(p, p.size // 3, n, n.size // 3, v, ni, T, nodes, tp, tn, codes, order, C.byref(sec), encoding)
    return dict(nodes=nodes.reshape(-1, 8), tri_pos=tp.reshape(T, 3, 4), tri_nor=tn.reshape(T, 3, 4),
                codes=codes, order=order, seconds=sec.value)


def build_sah(pos, nor, vidx, nidx, leaf=1):
    """The reference's own builder (script.js:285-513 + flattenForSSBO).  Same dict as build_lbvh, without codes."""
    p, v = _mesh(pos, vidx)
    n, ni = _mesh(nor, nidx)
    T = v.size // 3
    cap = 2 * T
    nodes = np.zeros(cap * 8, np.float32)
    tp, tn = np.zeros(T * 12, np.float32), np.zeros(T * 12, np.float32)
    order = np.zeros(T, np.uint32)
    sec, nn = C.c_double(), C.c_uint32()
    rc = lib().orc_build_sah(p, n, v, ni, T, leaf, nodes, cap, C.byref(nn), tp, tn, order, C.byref(sec))
    if rc != 0:
        raise RuntimeError(f"orc_build_sah failed: {rc}")
    return dict(nodes=nodes[:nn.value * 8].reshape(-1, 8).copy(), tri_pos=tp.reshape(T, 3, 4), tri_nor=tn.reshape(T, 3, 4),
                order=order, seconds=sec.value)


DEFAULT_SLIDERS = (1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


def make_uniforms(cam_pos=(0.0, 0.0, 5.0), yaw=0.0, pitch=0.0, sliders=DEFAULT_SLIDERS, width=1920, height=1080, time=0.0):
    cp = (C.c_double * 3)(*cam_pos)
    sl = (C.c_double * 9)(*sliders)
    out = np.zeros(44, np.float32)
    lib().orc_make_uniforms(cp, yaw, pitch, sl, width, height, time, out)
    return out


def render(nodes, tri_pos, tri_nor, uniforms, mode=MODE_PRIMARY, gi_samples=32, gi_bounces=3, soft_samples=4,
           max_stack=12, rows=None, n_threads=1, want=("rgba8", "hit_tri", "hit_t"), encoding=1, surface=None):
    """rows = (begin, end, step) or None for all. Returns (outputs dict, stats dict).
    surface: a Surface (material/texture extension) or None for the reference's constant albedo."""
    nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1)
    tp = np.ascontiguousarray(tri_pos, np.float32).reshape(-1)
    tn = np.ascontiguousarray(tri_nor, np.float32).reshape(-1)
    U = np.ascontiguousarray(uniforms, np.float32)
    w, h = int(U[36]), int(U[37])
    rb, re_, rs = rows if rows is not None else (0, h, 1)
    opts = RenderOpts(mode, gi_samples, gi_bounces, soft_samples, max_stack, rb, re_, rs, n_threads, encoding)
    out = {}
    if "rgba8" in want: out["rgba8"] = np.zeros((h, w, 4), np.uint8)
    if "hit_tri" in want: out["hit_tri"] = np.full((h, w), 0xFFFFFFFF, np.uint32)
    if "hit_t" in want: out["hit_t"] = np.zeros((h, w), np.float32)
    ptr = lambda k: out[k].ctypes.data_as(C.c_void_p) if k in out else None
    st = RenderStats()
    if surface is not None and surface.tri_uv.shape[0] != tp.size // 12:
        raise ValueError("surface.tri_uv must have one record per triangle")
    lib().orc_set_surface(C.addressof(surface.c) if surface is not None else None)
    try:
        rc = lib().orc_render(nodes, nodes.size // 8, tp, tn, tp.size // 12, U, C.byref(opts),
                              ptr("rgba8"), ptr("hit_tri"), ptr("hit_t"), C.byref(st))
    finally:
        lib().orc_set_surface(None)
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return out, st.as_dict()
