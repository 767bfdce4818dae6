"""raytracer_amd — host-side mirror of the reference's JS driver (web/script.js) over the
C ABI of librtcore.so (include/rtcore.h).  Names follow the reference: load_obj (loadObj
:164), normalize_mesh (:207), generate_test_scene (:639), build_lbvh (buildLBVH :285),
flatten_for_ssbo (:526), make_uniforms (udata packing :899-929), render (frame() :865).

All compute runs in the HIP library; there is no Python or CPU compute path here.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (FLAG_COUNT_WORK, FLAG_TEXTURED, LAYOUT_AOS32, LAYOUT_WIDE64, MODE_FRAG_ROOT, MODE_FRAG_WEB,  # noqa: F401
                   MODE_PRIMARY, MODE_PT, MODE_SHADOW, NODES_V1, NODES_V2)

_L = _abi.load()          # raises ImportError when the HIP extension is not built


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtcore error {code}: {msg}")
        self.code = code


class Mesh:
    """The loadObj() return contract (script.js:198-203) + texcoords / material ids."""

    def __init__(self, positions, normals, vert_idx, norm_idx, texcoords=None, tex_idx=None, material_ids=None, materials=()):
        self.positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self.vert_idx = np.ascontiguousarray(vert_idx, np.uint32).reshape(-1)
        self.norm_idx = np.ascontiguousarray(norm_idx, np.uint32).reshape(-1)
        self.texcoords = None if texcoords is None else np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
        self.tex_idx = None if tex_idx is None else np.ascontiguousarray(tex_idx, np.uint32).reshape(-1)
        self.material_ids = None if material_ids is None else np.ascontiguousarray(material_ids, np.int32)
        self.materials = list(materials)

    @property
    def n_tri(self):
        return self.vert_idx.size // 3


def _mesh_from_handle(h):
    try:
        v = _abi.MeshView()
        rc = _L.rt_mesh_view_get(h, C.byref(v))
        if rc != 0:
            raise RtError(rc, "rt_mesh_view_get")
        arr = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        mats = []
        for i in range(v.n_materials):
            m = _abi.Material()
            _L.rt_mesh_material(h, i, C.byref(m))
            unknown = {}
            for k in range(_L.rt_mesh_material_n_unknown(h, i)):
                key, val = C.c_char_p(), C.c_char_p()
                _L.rt_mesh_material_unknown(h, i, k, C.byref(key), C.byref(val))
                unknown[key.value.decode()] = val.value.decode()
            d = dict(name=m.name.decode(), diffuse=tuple(m.diffuse), emission=tuple(m.emission), dissolve=m.dissolve,
                     illum=m.illum, ambient=tuple(m.ambient), specular=tuple(m.specular), transmittance=tuple(m.transmittance),
                     shininess=m.shininess, ior=m.ior, roughness=m.roughness, metallic=m.metallic, sheen=m.sheen,
                     clearcoat_thickness=m.clearcoat_thickness, clearcoat_roughness=m.clearcoat_roughness,
                     anisotropy=m.anisotropy, anisotropy_rotation=m.anisotropy_rotation, unknown_parameter=unknown)
            for k, key in enumerate(_abi.MAP_KEYS):            # the thirteen texture names + options of material_t
                d[key] = m.texname[k].value.decode()
                d[key + "_opt"] = m.texopt[k].as_dict()
            mats.append(d)
        return Mesh(arr(v.positions, 3 * v.n_vert, np.float32), arr(v.normals, 3 * v.n_nor, np.float32),
                    arr(v.vert_idx, 3 * v.n_tri, np.uint32), arr(v.nor_idx, 3 * v.n_tri, np.uint32),
                    arr(v.texcoords, 2 * v.n_tex, np.float32), arr(v.tex_idx, 3 * v.n_tri, np.uint32),
                    arr(v.material_ids, v.n_tri, np.int32), mats)
    finally:
        _L.rt_mesh_free(h)


def load_obj(path, mtl_dir=None):
    """loadObj (script.js:164-204) with tiny_obj_loader parsing semantics."""
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = _L.rt_mesh_load_obj(str(path).encode(), None if mtl_dir is None else str(mtl_dir).encode(), C.byref(h), err, 512)
    if rc != 0:
        raise RtError(rc, err.value.decode())
    return _mesh_from_handle(h)


def parse_obj(obj_text, mtl_text=None):
    ob = obj_text.encode() if isinstance(obj_text, str) else bytes(obj_text)
    mb = None if mtl_text is None else (mtl_text.encode() if isinstance(mtl_text, str) else bytes(mtl_text))
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = _L.rt_mesh_parse_obj(ob, len(ob), mb, 0 if mb is None else len(mb), C.byref(h), err, 512)
    if rc != 0:
        raise RtError(rc, err.value.decode())
    return _mesh_from_handle(h)


def load_texture(path):
    """An image file as the RGB8 array rt_set_texture takes ([h, w, 3] uint8, row 0 = the file's first row).
    Binary PPM (P6, maxval 255) is read here; anything else (JPEG, PNG: what the reference's MTL files name) through PIL when
    it is installed.  The reference itself never reads a texture; this is the file side of RT_FLAG_TEXTURED."""
    with open(path, "rb") as f:
        head = f.read(2)
    if head == b"P6":
        data = open(path, "rb").read()
        tok, pos = [], 2
        while len(tok) < 3:                                  # width, height, maxval; '#' comments run to the end of the line
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            tok.append(int(data[pos:end]))
            pos = end
        w, h, mx = tok
        if mx != 255 or len(data) < pos + 1 + 3 * w * h:
            raise ValueError(f"{path}: only 8-bit P6 files of the stated size are read")
        return np.frombuffer(data, np.uint8, 3 * w * h, pos + 1).reshape(h, w, 3).copy()
    try:
        from PIL import Image
    except ImportError as e:
        raise RuntimeError(f"{path}: reading this image format needs PIL (only binary PPM is read without it)") from e
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def bind_obj_surface(ctx, mesh, texture_dir, loader=load_texture):
    """The surface of a mesh that load_obj read together with its MTL: texcoords + material ids (rt_set_surface), one material
    record per MTL material (Kd, and a texture slot where map_Kd names a file that exists under texture_dir), the images.
    -> the list of (map_Kd name, slot) actually bound.  Materials whose texture file is missing keep their Kd only."""
    import os
    ctx.set_surface(mesh)
    slots, recs = {}, []
    for m in mesh.materials:
        name = m.get("map_Kd", "") if isinstance(m, dict) else ""
        kd = m["diffuse"] if isinstance(m, dict) else m[0]
        slot = -1
        if name:
            f = os.path.join(texture_dir, name.replace("\\", "/"))
            if name in slots:
                slot = slots[name]
            elif os.path.isfile(f):
                slot = slots[name] = len(slots)
                ctx.set_texture(slot, loader(f))
        recs.append((kd, slot))
    ctx.set_materials(recs if recs else [((1.0, 1.0, 1.0), -1)])
    return sorted(slots.items(), key=lambda kv: kv[1])


def normalize_mesh(mesh):
    """normalizeMesh (script.js:207-227), in place on mesh.positions."""
    p = mesh.positions
    rc = _L.rt_normalize_mesh(p.ctypes.data_as(C.c_void_p), p.shape[0])
    if rc != 0:
        raise RtError(rc, "rt_normalize_mesh")
    return mesh


def generate_test_scene():
    """generateTestScene (script.js:639-675): the 2-triangle quad with +Z normals."""
    return Mesh(np.float32([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]), np.float32([[0, 0, 1]] * 4),
                np.uint32([0, 1, 2, 0, 2, 3]), np.uint32([0, 1, 2, 0, 2, 3]))


def make_uniforms(width, height, cam_pos=(0.0, 0.0, 5.0), yaw=0.0, pitch=0.0, scale=(1.0, 1.0, 1.0),
                  rot_deg=(0.0, 0.0, 0.0), translate=(0.0, 0.0, 0.0), time=0.0):
    """The 44 uniform floats of compute.wgsl:4-13 as frame() packs them (script.js:880-929)."""
    cam = _abi.Camera()
    _L.rt_camera_default(C.byref(cam), width, height)
    cam.pos[:] = cam_pos
    cam.yaw, cam.pitch = yaw, pitch
    cam.scale[:] = scale
    cam.rot_deg[:] = rot_deg
    cam.translate[:] = translate
    cam.time = time
    out = np.zeros(44, np.float32)
    rc = _L.rt_make_uniforms(C.byref(cam), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RtError(rc, "rt_make_uniforms")
    return out


def render_opts(mode=MODE_PT, gi_samples=32, gi_bounces=3, soft_samples=4, max_stack=12, layout=LAYOUT_WIDE64, flags=0,
                stripe_rows=0, stripe_rank=0, stripe_world=1):
    return _abi.RenderOpts(mode, gi_samples, gi_bounces, soft_samples, max_stack, layout, flags, stripe_rows, stripe_rank,
                           stripe_world)


def render_opts_for_mode(mode, **overrides):
    """The constants the shader of `mode` is written with (rt_render_opts_for_mode), then `overrides` by field name."""
    o = _abi.RenderOpts()
    rc = _L.rt_render_opts_for_mode(C.byref(o), mode)
    if rc != 0:
        raise RtError(rc, "rt_render_opts_for_mode")
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f"rt_render_opts has no field {k!r}")
        setattr(o, k, v)
    return o


def device_allocations():
    """hipMalloc + hipFree calls the library has made in this process (rt_debug_device_allocations)."""
    return int(_L.rt_debug_device_allocations())


def packed_rows(height, stripe_rows, rank, world):
    return int(_L.rt_packed_rows(height, stripe_rows, rank, world))


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU. Mirrors the page-lifetime state of the reference (device, buffers, pipelines)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        ids = (C.c_int * 1)(device)
        rc = _L.rt_create(C.byref(self._h), b"hip", ids, 1)
        if rc != 0:
            raise RtError(rc, _L.rt_last_error(None).decode())
        self.device = device

    def close(self):
        if self._h:
            _L.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, _L.rt_last_error(self._h).decode())

    # ---- hot path 1
    def set_mesh(self, mesh):
        self._chk(_L.rt_set_mesh(self._h, _vp(mesh.positions), mesh.positions.shape[0], _vp(mesh.normals),
                                 mesh.normals.shape[0], _vp(mesh.vert_idx), _vp(mesh.norm_idx), mesh.n_tri))

    def set_node_encoding(self, encoding):
        """NODES_V1 (reference, f32-valued indices, < 2^24 nodes) or NODES_V2 (u32 bit-casts) for the next build / import."""
        self._chk(_L.rt_set_node_encoding(self._h, encoding))

    def node_encoding(self):
        return int(_L.rt_get_node_encoding(self._h))

    def set_build_stage_timing(self, on=True):
        """Per-stage build times (ms_bounds ... ms_tree) on / off: off by default, the events between the launches cost ~2 us each."""
        self._chk(_L.rt_set_build_stage_timing(self._h, 1 if on else 0))

    def build_lbvh(self):
        st = _abi.BuildStats()
        self._chk(_L.rt_build_lbvh(self._h, C.byref(st)))
        return st.as_dict()

    def counts(self):
        t, n = C.c_uint32(), C.c_uint32()
        self._chk(_L.rt_get_counts(self._h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def bounds(self):
        b, i = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._chk(_L.rt_get_bounds(self._h, _vp(b), _vp(i)))
        return b, i

    def morton(self):
        c = np.zeros(self.counts()[0], np.uint32)
        self._chk(_L.rt_get_morton(self._h, _vp(c)))
        return c

    def tri_order(self):
        o = np.zeros(self.counts()[0], np.uint32)
        self._chk(_L.rt_get_tri_order(self._h, _vp(o)))
        return o

    def flatten_for_ssbo(self):
        """{triPos, triNor, bvhData} exactly as flattenForSSBO returns them (script.js:560)."""
        T, N = self.counts()
        nodes, tp, tn = np.zeros((N, 8), np.float32), np.zeros((T, 3, 4), np.float32), np.zeros((T, 3, 4), np.float32)
        self._chk(_L.rt_get_nodes(self._h, _vp(nodes)))
        self._chk(_L.rt_get_triangles(self._h, _vp(tp), _vp(tn)))
        return dict(triPos=tp, triNor=tn, bvhData=nodes)

    def set_bvh(self, nodes, tri_pos, tri_nor):
        nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 8)
        tp = np.ascontiguousarray(tri_pos, np.float32).reshape(-1, 12)
        tn = np.ascontiguousarray(tri_nor, np.float32).reshape(-1, 12)
        self._chk(_L.rt_set_bvh(self._h, _vp(nodes), nodes.shape[0], _vp(tp), _vp(tn), tp.shape[0]))

    def save_bvh(self, path):
        self._chk(_L.rt_save_bvh(self._h, str(path).encode()))

    def load_bvh(self, path):
        self._chk(_L.rt_load_bvh(self._h, str(path).encode()))

    # ---- material / texture shading (extension, rtcore.h)
    def set_surface(self, mesh):
        """Texcoords + per-face material ids of the mesh last given to set_mesh (before or after build_lbvh)."""
        if mesh.texcoords is None or mesh.tex_idx is None:
            raise ValueError("mesh has no texcoords")
        self._chk(_L.rt_set_surface(self._h, _vp(mesh.texcoords), mesh.texcoords.shape[0], _vp(mesh.tex_idx), _vp(mesh.material_ids)))

    def tri_uv(self):
        out = np.zeros((self.counts()[0], 8), np.float32)
        self._chk(_L.rt_get_tri_uv(self._h, _vp(out)))
        return out

    def set_tri_uv(self, tri_uv):
        a = np.ascontiguousarray(tri_uv, np.float32).reshape(-1, 8)
        self._chk(_L.rt_set_tri_uv(self._h, _vp(a), a.shape[0]))

    def set_materials(self, materials):
        """materials: sequence of (kd_rgb, texture_slot or -1/None)."""
        arr = (_abi.MaterialRec * max(len(materials), 1))()
        for i, (kd, slot) in enumerate(materials):
            arr[i].kd[:] = [float(x) for x in kd]
            arr[i].texture = -1 if slot is None else int(slot)
        self._chk(_L.rt_set_materials(self._h, arr, len(materials)))

    def set_texture(self, slot, rgb8):
        """rgb8: uint8 [H, W, 3], row 0 = top of the image."""
        a = np.ascontiguousarray(rgb8, np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("texture must be uint8 [H, W, 3]")
        self._chk(_L.rt_set_texture(self._h, slot, _vp(a), a.shape[1], a.shape[0]))

    def tri_uv_device(self):
        """-> (device address or 0, bytes) of the BVH-order surface records."""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(_L.rt_tri_uv_device(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def export_tri_uv_device(self, d_dst, n_tri):
        """Copy the surface records into caller-owned device memory (32 B x n_tri), e.g. the buffer RCCL broadcasts."""
        self._chk(_L.rt_tri_uv_export_device(self._h, C.c_void_p(d_dst), 32 * n_tri))

    def adopt_tri_uv_device(self, d_tri_uv, n_tri):
        self._chk(_L.rt_adopt_tri_uv_device(self._h, C.c_void_p(d_tri_uv), n_tri))

    # ---- hot path 2
    def render(self, uniforms, opts=None, want=("rgba8", "hit_tri", "hit_t")):
        """One frame() (script.js:865-945) into host arrays. Returns (outputs, stats)."""
        opts = opts or render_opts()
        U = np.ascontiguousarray(uniforms, np.float32)
        w, h = int(U[36]), int(U[37])
        out = {}
        if "rgba8" in want: out["rgba8"] = np.zeros((h, w, 4), np.uint8)
        if "hit_tri" in want: out["hit_tri"] = np.full((h, w), 0xFFFFFFFF, np.uint32)
        if "hit_t" in want: out["hit_t"] = np.zeros((h, w), np.float32)
        st = _abi.RenderStats()
        self._chk(_L.rt_render(self._h, _vp(U), C.byref(opts), _vp(out.get("rgba8")), _vp(out.get("hit_tri")),
                               _vp(out.get("hit_t")), C.byref(st)))
        return out, st.as_dict()

    def render_device(self, uniforms, opts, d_rgba8=0, d_hit_tri=0, d_hit_t=0, want_stats=True):
        """Outputs into caller-owned device memory (integer addresses, e.g. torch .data_ptr())."""
        U = np.ascontiguousarray(uniforms, np.float32)
        st = _abi.RenderStats()
        self._chk(_L.rt_render_device(self._h, _vp(U), C.byref(opts), C.c_void_p(d_rgba8 or None), C.c_void_p(d_hit_tri or None),
                                      C.c_void_p(d_hit_t or None), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def render_device_noalloc(self, uniforms, opts, d_rgba8=0, d_hit_tri=0, d_hit_t=0):
        """rt_render_device without any device allocation (static scratch, <= 3840x2160 pixels per call); asynchronous."""
        U = np.ascontiguousarray(uniforms, np.float32)
        self._chk(_L.rt_render_device_noalloc(self._h, _vp(U), C.byref(opts), C.c_void_p(d_rgba8 or None),
                                              C.c_void_p(d_hit_tri or None), C.c_void_p(d_hit_t or None)))

    def render_batch_device(self, uniforms, opts, d_rgba8=None, d_hit_tri=None, d_hit_t=None):
        """Several frames of one image shape in one pipeline (rt_render_batch_device).  uniforms: [n, 44]; each output
        argument: a sequence of n device addresses (0 = no such output for that frame) or None; asynchronous."""
        U = np.ascontiguousarray(uniforms, np.float32).reshape(-1, 44)
        n = U.shape[0]

        def ptrs(a):
            if a is None:
                return None
            if len(a) != n:
                raise ValueError("one device address per frame")
            return (C.c_void_p * n)(*[C.c_void_p(int(x) or None) for x in a])
        self._chk(_L.rt_render_batch_device(self._h, n, _vp(U), C.byref(opts), ptrs(d_rgba8), ptrs(d_hit_tri), ptrs(d_hit_t)))

    def batch_stats(self, n_frames):
        """Per-frame counters of the last render_batch_device call made with FLAG_COUNT_WORK (rt_batch_stats); synchronises."""
        arr = (_abi.RenderStats * n_frames)()
        self._chk(_L.rt_batch_stats(self._h, n_frames, arr))
        return [a.as_dict() for a in arr]

    def deinterleave_device(self, d_packed, d_image, width, height, elem_bytes, stripe_rows, world, rows_per_rank, stream=None):
        """stream: a hipStream_t (integer) of this GPU to launch on instead of the context stream."""
        if stream is None:
            self._chk(_L.rt_deinterleave_device(self._h, C.c_void_p(d_packed), C.c_void_p(d_image), width, height, elem_bytes,
                                                stripe_rows, world, rows_per_rank))
        else:
            self._chk(_L.rt_deinterleave_device_on(self._h, C.c_void_p(stream or None), C.c_void_p(d_packed), C.c_void_p(d_image),
                                                   width, height, elem_bytes, stripe_rows, world, rows_per_rank))

    def frame_rect(self, uniforms):
        """-> ((x0, y0, x1, y1), miss_rgba): the screen rectangle (inclusive) outside of which every pixel of the frame is a miss,
        and the value of a missed pixel (rt_frame_rect; the compute.wgsl modes)."""
        U = np.ascontiguousarray(uniforms, np.float32)
        r, miss = (C.c_int32 * 4)(), C.c_uint32()
        self._chk(_L.rt_frame_rect(self._h, _vp(U), C.byref(r), C.byref(miss)))
        return (r[0], r[1], r[2], r[3]), miss.value

    def assemble_device(self, d_windows, d_image, width, height, stripe_rows, world, col0, n_cols, group0, n_rows, rank_stride, fill, stream=None):
        """The full image from the ranks' gathered windows (rt_assemble_device_on); stream: a hipStream_t (integer) or None."""
        self._chk(_L.rt_assemble_device_on(self._h, C.c_void_p(stream or None), C.c_void_p(d_windows or None), C.c_void_p(d_image), width, height,
                                           stripe_rows, world, col0, n_cols, group0, n_rows, rank_stride, fill))

    # ---- multi-GPU plumbing
    def bvh_view(self):
        v = _abi.BvhView()
        self._chk(_L.rt_bvh_view_get(self._h, C.byref(v)))
        return v

    def export_bvh_device(self, d_nodes, d_tri_pos, d_tri_nor, n_nodes, n_tri):
        v = _abi.BvhView(d_nodes, 32 * n_nodes, d_tri_pos, 48 * n_tri, d_tri_nor, 48 * n_tri, n_nodes, n_tri)
        self._chk(_L.rt_bvh_export_device(self._h, C.byref(v)))

    def adopt_bvh_device(self, d_nodes, d_tri_pos, d_tri_nor, n_nodes, n_tri):
        v = _abi.BvhView(d_nodes, 32 * n_nodes, d_tri_pos, 48 * n_tri, d_tri_nor, 48 * n_tri, n_nodes, n_tri)
        self._chk(_L.rt_bvh_adopt_device(self._h, C.byref(v)))

    def set_coherent_scheduling(self, on):
        """Frame-coherent scheduling hints (rtcore.h): on by default; results never depend on it."""
        self._chk(_L.rt_set_coherent_scheduling(self._h, 1 if on else 0))

    def set_frames_in_flight(self, n):
        """1 (default): frames run in order on stream(); 2: asynchronous frames alternate between two frame streams."""
        self._chk(_L.rt_set_frames_in_flight(self._h, n))

    def frames_join(self, keep_latest=0):
        """stream() waits for the asynchronous frames issued so far (all but the most recent one with keep_latest=1)."""
        self._chk(_L.rt_frames_join(self._h, keep_latest))

    def synchronize(self):
        self._chk(_L.rt_synchronize(self._h))

    def stream(self):
        return _L.rt_stream(self._h)

    def profile_begin(self):
        self._chk(_L.rt_profile_begin(self._h))

    def profile_end_phases(self):
        """-> (ms_frame_avg, ms_traverse_avg, ms_shade_avg, n): as profile_end, the traversal time by launch (rt_profile_end_phases)."""
        a, b, c, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint32()
        self._chk(_L.rt_profile_end_phases(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def profile_end(self):
        """-> (ms_frame_avg, ms_traverse_avg, n_frames) over the frames launched since profile_begin."""
        a, b, n = C.c_float(), C.c_float(), C.c_uint32()
        self._chk(_L.rt_profile_end(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value
