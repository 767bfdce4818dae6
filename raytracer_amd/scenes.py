"""Closed-form stand-in scenes for the meshes the reference does not ship:
`web/bunny.obj` is absent and `web/dragon.obj` is a Git-LFS pointer.

  S70k  icosphere, 6 subdivisions = 81 920 triangles, radius 1 + 0.15 sin(7θ) sin(5φ)
  S1M   icosphere, 8 subdivisions = 1 310 720 triangles, + 0.02 sin(40θ) sin(40φ)
  lattice(K)  K copies of a mesh on a ceil(cbrt K)^3 lattice, spacing 1.1, each rotated about Y
              by 2π·frac(k·0.618034), flattened (BASELINE cfg 5)

Every stand-in also carries what an OBJ/MTL pair would (tinyobj attrib_t::texcoords, mesh_t::material_ids):
spherical texcoords u = theta/2pi + 1/2, v = 1 - phi/pi per vertex, three materials in latitude bands
(STANDIN_MATERIALS) and closed-form textures (procedural_texture) in place of the reference's JPG/PNG files under
models/, which no GPU run can read.

Host-side numpy only (scene *input*, not the hot path); seed-free and deterministic.
"""
import numpy as np

from . import Mesh


def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    return v / np.linalg.norm(v, axis=1, keepdims=True), f


def _subdivide(v, f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * (v.shape[0] + 1) + e[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    a, b = uniq // (v.shape[0] + 1), uniq % (v.shape[0] + 1)
    mid = v[a] + v[b]
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    n0 = v.shape[0]
    m = inv.reshape(3, -1) + n0            # midpoints of edges (01, 12, 20) per face
    v2 = np.concatenate([v, mid], axis=0)
    f2 = np.concatenate([np.stack([f[:, 0], m[0], m[2]], 1), np.stack([f[:, 1], m[1], m[0]], 1),
                         np.stack([f[:, 2], m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)], axis=0)
    return v2, f2


def icosphere(levels, fine_ripple=False):
    v, f = _icosahedron()
    for _ in range(levels):
        v, f = _subdivide(v, f)
    theta = np.arctan2(v[:, 2], v[:, 0])
    phi = np.arccos(np.clip(v[:, 1], -1.0, 1.0))
    r = 1.0 + 0.15 * np.sin(7.0 * theta) * np.sin(5.0 * phi)
    if fine_ripple:
        r = r + 0.02 * np.sin(40.0 * theta) * np.sin(40.0 * phi)
    p = v * r[:, None]
    # smooth normals: area-weighted face normals accumulated per vertex
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
    idx = f.astype(np.uint32).reshape(-1)
    uv = np.stack([theta / (2.0 * np.pi) + 0.5, 1.0 - phi / np.pi], axis=1)
    # materials in latitude bands of the face centroid: 0 textured, 1 plain Kd, 2 tinted + second texture
    band = np.floor(np.arccos(np.clip(v[f].mean(axis=1)[:, 1], -1.0, 1.0)) / np.pi * 9.0).astype(np.int64)
    mat = (np.clip(band, 0, 8) % 3).astype(np.int32)
    return Mesh(p.astype(np.float32), vn.astype(np.float32), idx, idx.copy(), texcoords=uv.astype(np.float32), tex_idx=idx.copy(),
                material_ids=mat, materials=STANDIN_MATERIALS)


# (Kd, texture slot): what models/*.mtl would give through tinyobj material_t::diffuse / diffuse_texname
STANDIN_MATERIALS = [((1.0, 1.0, 1.0), 0), ((0.9, 0.5, 0.2), -1), ((0.6, 0.9, 1.0), 1)]


def procedural_texture(size=2048, variant=0):
    """Closed-form RGB8 image [size, size, 3] (row 0 = top): colour ramps with a 32-texel checker in the third
    channel; variant 1 swaps the channels and uses 8-texel cells."""
    y, x = np.mgrid[0:size, 0:size]
    cell = 32 if variant == 0 else 8
    ramp_x = (x * 255 // max(size - 1, 1)).astype(np.uint8)
    ramp_y = (y * 255 // max(size - 1, 1)).astype(np.uint8)
    check = np.where(((x // cell) + (y // cell)) % 2 == 0, 255, 64).astype(np.uint8)
    ch = [ramp_x, ramp_y, check] if variant == 0 else [check, ramp_x, ramp_y]
    return np.ascontiguousarray(np.stack(ch, axis=2))


def bind_standin_surface(ctx, mesh, tex_size=2048):
    """Upload a stand-in mesh's texcoords / material ids, its materials and the two closed-form textures."""
    ctx.set_surface(mesh)
    ctx.set_materials(mesh.materials)
    ctx.set_texture(0, procedural_texture(tex_size, 0))
    ctx.set_texture(1, procedural_texture(max(tex_size // 2, 1), 1))


def s70k():
    return icosphere(6)


def s1m():
    return icosphere(8, fine_ripple=True)


def lattice(mesh, copies):
    side = int(np.ceil(copies ** (1.0 / 3.0) - 1e-9))
    P, N, VI, NI = [], [], [], []
    nv, nn = mesh.positions.shape[0], mesh.normals.shape[0]
    textured = mesh.texcoords is not None and mesh.tex_idx is not None
    for k in range(copies):
        ang = 2.0 * np.pi * ((k * 0.618034) % 1.0)
        c, s = np.cos(ang), np.sin(ang)
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
        off = 1.1 * np.array([k % side, (k // side) % side, k // (side * side)], np.float64)
        P.append((mesh.positions.astype(np.float64) @ R.T + off).astype(np.float32))
        N.append((mesh.normals.astype(np.float64) @ R.T).astype(np.float32))
        VI.append(mesh.vert_idx + np.uint32(k * nv))
        NI.append(mesh.norm_idx + np.uint32(k * nn))
    if not textured:
        return Mesh(np.concatenate(P), np.concatenate(N), np.concatenate(VI), np.concatenate(NI))
    # texcoords are shared by all copies (indices are not offset); material ids repeat per copy
    return Mesh(np.concatenate(P), np.concatenate(N), np.concatenate(VI), np.concatenate(NI), texcoords=mesh.texcoords,
                tex_idx=np.tile(mesh.tex_idx, copies),
                material_ids=None if mesh.material_ids is None else np.tile(mesh.material_ids, copies), materials=mesh.materials)
