"""ctypes declarations for librtcore.so — the exact entry points of include/rtcore.h.

No torch types cross this boundary; device buffers are passed as integer addresses.
There is no CPU fallback: if the HIP library is missing, importing this module fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTCORE_LIBRARY") or os.path.join(_HERE, "lib", "librtcore.so")   # RTCORE_LIBRARY: another build of the same ABI (A/B experiments)

RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_IO, RT_ERR_LIMIT, RT_ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6
MODE_PRIMARY, MODE_SHADOW, MODE_PT = 0, 1, 2
MODE_FRAG_WEB, MODE_FRAG_ROOT = 3, 4      # the older fragment shaders: web/shader.wgsl, shader.wgsl
LAYOUT_WIDE64, LAYOUT_AOS32 = 0, 1
FLAG_COUNT_WORK = 1
FLAG_TEXTURED = 2
MAX_TEXTURES = 64
NODES_V1, NODES_V2 = 1, 2

# every symbol include/rtcore.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_abi_version", "rt_set_build_stage_timing",
    "rt_mesh_load_obj", "rt_mesh_parse_obj", "rt_mesh_view_get", "rt_mesh_material", "rt_mesh_material_n_unknown", "rt_mesh_material_unknown", "rt_mesh_free",
    "rt_normalize_mesh", "rt_camera_default", "rt_make_uniforms",
    "rt_set_mesh", "rt_build_lbvh", "rt_get_bounds", "rt_get_morton", "rt_get_tri_order", "rt_get_counts",
    "rt_get_nodes", "rt_get_triangles", "rt_set_bvh", "rt_save_bvh", "rt_load_bvh",
    "rt_render_opts_default", "rt_render_opts_for_mode", "rt_render", "rt_render_device", "rt_packed_rows", "rt_deinterleave_device", "rt_deinterleave_device_on",
    "rt_bvh_view_get", "rt_bvh_export_device", "rt_bvh_adopt_device", "rt_synchronize", "rt_stream",
    "rt_profile_begin", "rt_profile_end", "rt_set_node_encoding", "rt_get_node_encoding",
    "rt_set_surface", "rt_get_tri_uv", "rt_set_tri_uv", "rt_set_materials", "rt_set_texture", "rt_tri_uv_device",
    "rt_adopt_tri_uv_device", "rt_tri_uv_export_device", "rt_set_frames_in_flight", "rt_frames_join", "rt_set_coherent_scheduling",
    "rt_render_device_noalloc",
    "rt_render_batch_device", "rt_frame_rect", "rt_assemble_device_on", "rt_batch_stats", "rt_debug_device_allocations", "rt_profile_end_phases",
]


class MeshView(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("n_vert", C.c_uint32),
                ("normals", C.POINTER(C.c_float)), ("n_nor", C.c_uint32),
                ("texcoords", C.POINTER(C.c_float)), ("n_tex", C.c_uint32),
                ("vert_idx", C.POINTER(C.c_uint32)), ("nor_idx", C.POINTER(C.c_uint32)), ("tex_idx", C.POINTER(C.c_uint32)),
                ("material_ids", C.POINTER(C.c_int32)), ("n_tri", C.c_uint32), ("n_materials", C.c_uint32)]


N_MAPS = 13
MAP_KEYS = ("map_Ka", "map_Kd", "map_Ks", "map_Ns", "map_Bump", "disp", "map_d", "refl", "map_Pr", "map_Pm", "map_Ps", "map_Ke", "norm")   # RT_MAP_* order


class TexOpt(C.Structure):
    _fields_ = [("type", C.c_int32), ("sharpness", C.c_float), ("brightness", C.c_float), ("contrast", C.c_float),
                ("origin_offset", C.c_float * 3), ("scale", C.c_float * 3), ("turbulence", C.c_float * 3),
                ("texture_resolution", C.c_int32), ("clamp", C.c_int32), ("blendu", C.c_int32), ("blendv", C.c_int32),
                ("bump_multiplier", C.c_float), ("imfchan", C.c_char), ("colorspace", C.c_char * 31)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        for k in ("origin_offset", "scale", "turbulence"):
            d[k] = tuple(d[k])
        d["imfchan"] = d["imfchan"].decode()
        d["colorspace"] = d["colorspace"].decode()
        return d


class Material(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("diffuse", C.c_float * 3), ("emission", C.c_float * 3),
                ("dissolve", C.c_float), ("illum", C.c_int), ("diffuse_texname", C.c_char * 256),
                ("bump_texname", C.c_char * 256),
                ("ambient", C.c_float * 3), ("specular", C.c_float * 3), ("transmittance", C.c_float * 3),
                ("shininess", C.c_float), ("ior", C.c_float)] + \
               [(n, C.c_float) for n in ("roughness", "metallic", "sheen", "clearcoat_thickness", "clearcoat_roughness",
                                         "anisotropy", "anisotropy_rotation")] + \
               [("texname", (C.c_char * 256) * N_MAPS), ("texopt", TexOpt * N_MAPS)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("yaw", C.c_double), ("pitch", C.c_double), ("scale", C.c_double * 3),
                ("rot_deg", C.c_double * 3), ("translate", C.c_double * 3), ("width", C.c_uint32), ("height", C.c_uint32),
                ("time", C.c_float)]


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ms_bounds", "ms_morton", "ms_sort", "ms_tree", "ms_refit", "ms_pack", "ms_total")] + \
               [("n_tri", C.c_uint32), ("n_nodes", C.c_uint32), ("algorithmic_bytes", C.c_uint64), ("refit_fallbacks", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RenderOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mode", "gi_samples", "gi_bounces", "soft_samples", "max_stack", "layout", "flags",
                                         "stripe_rows", "stripe_rank", "stripe_world")]


class RenderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests",
                                          "hit_updates", "pixels", "stack_drops")] + [("ms_kernel", C.c_float), ("ms_traverse", C.c_float),
                                                                                   ("pixels_traced", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MaterialRec(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("texture", C.c_int32)]


class BvhView(C.Structure):
    _fields_ = [("d_nodes", C.c_void_p), ("nodes_bytes", C.c_uint64), ("d_tri_pos", C.c_void_p), ("tri_pos_bytes", C.c_uint64),
                ("d_tri_nor", C.c_void_p), ("tri_nor_bytes", C.c_uint64), ("n_nodes", C.c_uint32), ("n_tri", C.c_uint32)]


def _share_hip_runtime_with_torch():
    """One process, one HIP runtime.  librtcore.so needs libamdhip64 (the system one under /opt/rocm by
    default); a PyTorch-ROCm wheel bundles its own copy, and whichever copy is loaded second finds no GPU
    ("No HIP GPUs are available").  dist.py and bench.py use torch for device buffers and RCCL next to this
    library, so when a torch wheel with a bundled runtime is installed, that copy is loaded first and
    librtcore.so binds to it (same SONAME).  RTCORE_HIP_RUNTIME=system skips this (torch-free hosts)."""
    if os.environ.get("RTCORE_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()). "
                          "raytracer_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32p = C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_float)
    u32p, cp = C.POINTER(C.c_uint32), C.c_char_p
    sig = {
        "rt_create": (C.c_int, [C.POINTER(vp), cp, C.POINTER(C.c_int), C.c_int]),
        "rt_destroy": (None, [vp]),
        "rt_last_error": (cp, [vp]),
        "rt_debug_device_allocations": (C.c_uint64, []),
        "rt_abi_version": (C.c_int, []),
        "rt_set_build_stage_timing": (C.c_int, [vp, C.c_int]),
        "rt_mesh_load_obj": (C.c_int, [cp, cp, C.POINTER(vp), C.c_char_p, C.c_size_t]),
        "rt_mesh_parse_obj": (C.c_int, [cp, C.c_size_t, cp, C.c_size_t, C.POINTER(vp), C.c_char_p, C.c_size_t]),
        "rt_mesh_view_get": (C.c_int, [vp, C.POINTER(MeshView)]),
        "rt_mesh_material": (C.c_int, [vp, u32, C.POINTER(Material)]),
        "rt_mesh_material_n_unknown": (u32, [vp, u32]),
        "rt_mesh_material_unknown": (C.c_int, [vp, u32, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
        "rt_mesh_free": (None, [vp]),
        "rt_normalize_mesh": (C.c_int, [vp, u32]),
        "rt_camera_default": (None, [C.POINTER(Camera), u32, u32]),
        "rt_make_uniforms": (C.c_int, [C.POINTER(Camera), vp]),
        "rt_set_mesh": (C.c_int, [vp, vp, u32, vp, u32, vp, vp, u32]),
        "rt_build_lbvh": (C.c_int, [vp, C.POINTER(BuildStats)]),
        "rt_get_bounds": (C.c_int, [vp, vp, vp]),
        "rt_get_morton": (C.c_int, [vp, vp]),
        "rt_get_tri_order": (C.c_int, [vp, vp]),
        "rt_get_counts": (C.c_int, [vp, u32p, u32p]),
        "rt_get_nodes": (C.c_int, [vp, vp]),
        "rt_get_triangles": (C.c_int, [vp, vp, vp]),
        "rt_set_bvh": (C.c_int, [vp, vp, u32, vp, vp, u32]),
        "rt_save_bvh": (C.c_int, [vp, cp]),
        "rt_load_bvh": (C.c_int, [vp, cp]),
        "rt_render_opts_default": (None, [C.POINTER(RenderOpts)]),
        "rt_render_opts_for_mode": (C.c_int, [C.POINTER(RenderOpts), C.c_int]),
        "rt_render": (C.c_int, [vp, vp, C.POINTER(RenderOpts), vp, vp, vp, C.POINTER(RenderStats)]),
        "rt_render_device": (C.c_int, [vp, vp, C.POINTER(RenderOpts), vp, vp, vp, C.POINTER(RenderStats)]),
        "rt_render_device_noalloc": (C.c_int, [vp, vp, C.POINTER(RenderOpts), vp, vp, vp]),
        "rt_render_batch_device": (C.c_int, [vp, C.c_uint32, vp, C.POINTER(RenderOpts), vp, vp, vp]),
        "rt_profile_end_phases": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
        "rt_batch_stats": (C.c_int, [vp, C.c_uint32, C.POINTER(RenderStats)]),
        "rt_packed_rows": (u32, [u32, i32, i32, i32]),
        "rt_deinterleave_device": (C.c_int, [vp, vp, vp, u32, u32, u32, i32, i32, u32]),
        "rt_deinterleave_device_on": (C.c_int, [vp, vp, vp, vp, u32, u32, u32, i32, i32, u32]),
        "rt_frame_rect": (C.c_int, [vp, vp, C.POINTER(C.c_int32 * 4), C.POINTER(u32)]),
        "rt_assemble_device_on": (C.c_int, [vp, vp, vp, vp, u32, u32, i32, i32, u32, u32, u32, u32, C.c_uint64, u32]),
        "rt_bvh_view_get": (C.c_int, [vp, C.POINTER(BvhView)]),
        "rt_bvh_export_device": (C.c_int, [vp, C.POINTER(BvhView)]),
        "rt_bvh_adopt_device": (C.c_int, [vp, C.POINTER(BvhView)]),
        "rt_synchronize": (C.c_int, [vp]),
        "rt_stream": (vp, [vp]),
        "rt_set_node_encoding": (C.c_int, [vp, C.c_int]),
        "rt_get_node_encoding": (C.c_int, [vp]),
        "rt_profile_begin": (C.c_int, [vp]),
        "rt_profile_end": (C.c_int, [vp, vp, vp, vp]),
        "rt_set_surface": (C.c_int, [vp, vp, u32, vp, vp]),
        "rt_get_tri_uv": (C.c_int, [vp, vp]),
        "rt_set_tri_uv": (C.c_int, [vp, vp, u32]),
        "rt_set_materials": (C.c_int, [vp, C.POINTER(MaterialRec), u32]),
        "rt_set_texture": (C.c_int, [vp, u32, vp, u32, u32]),
        "rt_tri_uv_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "rt_adopt_tri_uv_device": (C.c_int, [vp, vp, u32]),
        "rt_tri_uv_export_device": (C.c_int, [vp, vp, C.c_uint64]),
        "rt_set_frames_in_flight": (C.c_int, [vp, C.c_int]),
        "rt_frames_join": (C.c_int, [vp, C.c_int]),
        "rt_set_coherent_scheduling": (C.c_int, [vp, C.c_int]),
    }
    for name in SYMBOLS:
        fn = getattr(L, name)          # AttributeError here = the library does not export a declared symbol
        fn.restype, fn.argtypes = sig[name]
    return L
