#!/usr/bin/env python3
"""Float64 working of two known answers that pin the CPU oracle (tests/test_oracle_known_answers.py).

Everything here is restated from the WGSL TEXT of web/compute.wgsl (cited per function) in plain float64 numpy, on a
two-triangle scene whose flat buffers are written down by hand - no code of oracle/ or of the library is used.  Only the
PRNG seed pipeline is evaluated in f32, because its results are TRUNCATED to integers there (rand2, :145-150) and a
float64 evaluation would land on other integers; from the jitters on, everything is float64.

Scene (object space = world space, model = identity):
  triangle 0 "O": an occluder that hovers above the receiver, its FRONT face looking down at it (a small one in case A,
                  a large one in case B);
  triangle 1 "R": a large receiver, tilted against all three axes, its front face looking up at the camera.
  tree: node 0 = root {child0 = 1, child1 = 2}, node 1 = leaf {O}, node 2 = leaf {R}  (v1 records: min.xyz, child0 | tri_start,
        max.xyz, child1 | -count; web/script.js:518-529).
  shadowRay (:166-196) only ever follows child0 - root, then node 1 - so only O can occlude; that is the reference's
  behaviour and it is restated as it stands.

Case A (SURVEY §8c "soft shadow"): one pixel, primary ray + computeDirectRadiance_OS_fast with SOFT_SAMPLES = 4 and
  GI_SAMPLES = 32 (a 6 x 6 grid, :217-224): the four sample directions, their `visible` flags, which of them hit O, occ,
  shadowFactor, the pixel's colour.
Case B: the same pixel under another light through computeLighting_optimized (:344-410) with GI_SAMPLES = 2 (a 2 x 1 grid),
  GI_BOUNCES = 1: both bounce directions, where they land on O, their direct terms, `indirect`, the pixel's colour.

Run it to print the numbers; the test imports `case_a()` / `case_b()` and compares the oracle's pixel with them."""
import json
import math

import numpy as np

F = np.float32
TWO_PI = 6.283185307179586
ALBEDO = np.array([1.0, 0.84, 0.0])


# ---------------------------------------------------------------- integers and f32 seeds (compute.wgsl:135-150)
def hash_u32(x):
    x &= 0xFFFFFFFF
    x = (x ^ 61) ^ (x >> 16)
    x = (x + (x << 3)) & 0xFFFFFFFF
    x = x ^ (x >> 4)
    x = (x * 0x27D4EB2D) & 0xFFFFFFFF
    return x ^ (x >> 15)


def to_u32(f):
    """WGSL u32(f32): truncation, saturating at both ends."""
    f = float(f)
    if not f > 0.0:
        return 0
    return 0xFFFFFFFF if f >= 4294967296.0 else int(f)


def rand2(sx, sy):
    """:145-150, every operation rounded to f32."""
    xi = to_u32(F(F(F(sx) * F(43758.0)) + F(24521.0)))
    yi = to_u32(F(F(F(sy) * F(199999.0)) + F(1423.0)))
    mix = xi ^ (yi >> 6) ^ ((yi << 11) & 0xFFFFFFFF)
    return float(F(F(hash_u32(mix)) * F(1.0 / 4294967296.0)))


TIME_H = F(hash_u32(0))          # cs_main passes f32(hash_u32(u32(uniforms.time))), time = 0 (:444)


# ---------------------------------------------------------------- scene
O_A = np.array([[-0.9, -0.7, 1.55], [0.1, 1.1, 1.75], [1.0, -0.6, 1.45]])     # case A: small, two of the four shadow samples pass it
O_B = np.array([[-3.0, -3.0, 1.5], [0.0, 3.5, 1.8], [3.5, -2.5, 1.4]])        # case B: large, both bounce rays land on it
R = np.array([[-2.0, -1.5, 0.3], [2.0, -1.2, -0.4], [0.2, 2.2, 0.5]])         # front face looks up; the occluders' front faces (e1 x e2) look down
TRIS, N_GEO, NORMALS = [], [], []


def unit(v):
    return v / np.linalg.norm(v)


def use_scene(occluder):
    TRIS[:] = [occluder, R]
    N_GEO[:] = [unit(np.cross(t[1] - t[0], t[2] - t[0])) for t in TRIS]
    NORMALS[:] = [np.tile(n, (3, 1)) for n in N_GEO]                           # flat shading: vertex normal = face normal


def flat_buffers():
    """(nodes [3, 8], tri_pos [2, 3, 4], tri_nor [2, 3, 4]) as f32, the layout of flattenForSSBO (script.js:515-545)."""
    P = np.stack(TRIS).astype(np.float32)
    mn, mx = P.min(axis=1), P.max(axis=1)
    nodes = np.zeros((3, 8), np.float32)
    nodes[0] = [*np.minimum(mn[0], mn[1]), 1, *np.maximum(mx[0], mx[1]), 2]
    nodes[1] = [*mn[0], 0, *mx[0], -1]
    nodes[2] = [*mn[1], 1, *mx[1], -1]
    tp = np.zeros((2, 3, 4), np.float32); tp[:, :, :3] = P
    tn = np.zeros((2, 3, 4), np.float32); tn[:, :, :3] = np.stack(NORMALS).astype(np.float32)
    return nodes, tp, tn


def boxes():
    n, _, _ = flat_buffers()
    return n.astype(np.float64)


W, H = 8, 8
CAM = np.array([0.0, 0.0, 5.0])
CAM_RIGHT, CAM_UP, CAM_FWD = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, -2.5])


def uniforms(light):
    U = np.zeros(44, np.float32)
    U[0:3] = CAM
    U[4:7] = light; U[7] = 1.0
    U[8:11], U[12:15], U[16:19] = CAM_RIGHT, CAM_UP, CAM_FWD
    U[20:36] = np.eye(4, dtype=np.float32).reshape(-1)
    U[36], U[37], U[38] = W, H, 0.0
    return U


# ---------------------------------------------------------------- geometry (compute.wgsl:90-131)
def slab(ro, inv, mn, mx):
    t0, t1 = (mn - ro) * inv, (mx - ro) * inv
    return np.minimum(t0, t1).max(), np.maximum(t0, t1).min()


def tri_hit(ro, rd, tri):
    v0, v1, v2 = tri
    e1, e2 = v1 - v0, v2 - v0
    p = np.cross(rd, e2)
    det = e1 @ p
    if not det > 1e-10:
        return None
    s = ro - v0
    u = (s @ p) / det
    q = np.cross(s, e1)
    v = (rd @ q) / det
    t = (e2 @ q) / det
    return (t, u, v) if (u >= 0 and v >= 0 and u + v <= 1 and t > 1e-10) else None


def closest(ro, rd):
    """traverseBVH_optimized (:260-341) on the 3-node tree: both leaves in near-first order, back faces skipped (:282)."""
    B = boxes()
    inv = 1.0 / rd
    best_t, best = 1e20, None
    order = []
    b0, b1 = slab(ro, inv, B[1, 0:3], B[1, 4:7]), slab(ro, inv, B[2, 0:3], B[2, 4:7])
    rb = slab(ro, inv, B[0, 0:3], B[0, 4:7])
    if not (rb[1] >= max(rb[0], 0.0) and rb[0] <= best_t):
        return best_t, None, None
    h0, h1 = b0[1] >= max(b0[0], 0.0), b1[1] >= max(b1[0], 0.0)
    if h0 and h1:
        order = [0, 1] if b0[0] <= b1[0] else [1, 0]
    elif h0:
        order = [0]
    elif h1:
        order = [1]
    for k in order:
        bk = (b0, b1)[k]
        if not (bk[1] >= max(bk[0], 0.0) and bk[0] <= best_t):      # re-tested when popped (:268)
            continue
        tri = TRIS[k]
        if np.cross(tri[1] - tri[0], tri[2] - tri[0]) @ rd > 0.0:
            continue
        h = tri_hit(ro, rd, tri)
        if h and h[0] < best_t:
            best_t, best = h[0], k
    n = None
    if best is not None:
        n = unit(NORMALS[best][0])                                     # the three vertex normals are equal
    return best_t, n, best


def shadow_ray(ro, rd, max_dist):
    """:166-196 - root, then child0 = node 1 = leaf {O}; nothing else is ever visited."""
    B = boxes()
    inv = 1.0 / rd
    for node in (0, 1):
        tn, tf = slab(ro, inv, B[node, 0:3], B[node, 4:7])
        if not (tf >= max(tn, 0.0) and tn <= max_dist):
            return False
    h = tri_hit(ro, rd, TRIS[0])
    return bool(h and h[0] < max_dist)


def direct(hit, n, light, seed_id, gi_samples, soft_samples, log=None):
    """computeDirectRadiance_OS_fast (:199-256)."""
    to_light = light - hit
    L = unit(to_light)
    diff = max(n @ L, 0.0)
    if diff <= 0.0:
        return 0.0
    up = np.array([1.0, 0.0, 0.0]) if abs(n[1]) > 0.99 else np.array([0.0, 1.0, 0.0])
    tangent = unit(np.cross(up, n))
    bitangent = np.cross(n, tangent)
    ro2 = hit + 0.001 * n
    light_dist = np.linalg.norm(to_light)
    cols = int(math.ceil(math.sqrt(gi_samples)))
    rows = (gi_samples + cols - 1) // cols
    occ = 0
    for i in range(soft_samples):
        sx, sy = i % cols, i // cols
        j1 = rand2(F(F(seed_id + 17) + TIME_H), F(F(i) + F(0.37)))
        j2 = rand2(F(F(seed_id + 91) + TIME_H), F(F(i) + F(1.13)))
        s1, s2 = (sx + j1) / cols, (sy + j2) / rows
        theta, r = TWO_PI * s1, math.sqrt(s2)
        d = unit(tangent * (r * math.cos(theta)) + bitangent * (r * math.sin(theta)) + n * math.sqrt(max(0.0, 1.0 - s2)))
        visible = bool(d @ L > 0.5)
        hit_o = shadow_ray(ro2, d, light_dist - 0.001)
        occ += 1 if (visible and hit_o) else 0
        if log is not None:
            log.append(dict(i=i, jitter=[j1, j2], s=[s1, s2], dir=d.tolist(), cos_to_light=float(d @ L), visible=visible, hits_occluder=hit_o))
    return diff * (1.0 - occ / soft_samples), occ, diff


def sample_hemisphere(n, s1, s2):
    """sampleHemisphere_importance (:153-163)."""
    theta, r = TWO_PI * s1, math.sqrt(s2)
    up = np.array([0.0, 1.0, 0.0]) if abs(n[1]) < 0.999 else np.array([1.0, 0.0, 0.0])
    T = unit(np.cross(up, n))
    Bv = np.cross(n, T)
    return unit(T * (r * math.cos(theta)) + Bv * (r * math.sin(theta)) + n * math.sqrt(max(0.0, 1.0 - s2)))


def primary(gx, gy):
    """cs_main (:414-446)."""
    u, v = (gx + 0.5) / W, (gy + 0.5) / H
    sx, sy = 2.0 * u - 1.0, 2.0 * v - 1.0
    rd = unit(CAM_RIGHT * sx + CAM_UP * sy + CAM_FWD)
    ix, iy = int(min(max(u * W, 0.0), W - 1)), int(min(max(v * H, 0.0), H - 1))
    return rd, iy * W + ix


def rgba8(col):
    g = np.power(np.maximum(col, 0.0), 1.0 / 2.2)
    return [int(math.floor(min(max(x, 0.0), 1.0) * 255.0 + 0.5)) for x in g] + [255]


PIXEL = (4, 3)
LIGHT_A = np.array([1.5, 2.0, 4.0])
LIGHT_B = np.array([2.6, 0.4, 0.9])


def case_a():
    use_scene(O_A)
    rd, pixel_index = primary(*PIXEL)
    t, n, tri = closest(CAM, rd)
    hit = CAM + rd * t
    log = []
    d, occ, diff = direct(hit, n, LIGHT_A, pixel_index, 32, 4, log)
    col = ALBEDO * d                                                   # primary + shadow rays only: no indirect term
    return dict(pixel=PIXEL, pixel_index=pixel_index, hit_tri=tri, hit_t=t, hit_pos=hit.tolist(), normal=n.tolist(), diff=diff,
                samples=log, occ=occ, shadow_factor=1.0 - occ / 4, direct=d, rgba8=rgba8(col))


def case_b(gi_samples=2, gi_bounces=1):
    use_scene(O_B)
    rd, pixel_index = primary(*PIXEL)
    t, n, tri = closest(CAM, rd)
    hit = CAM + rd * t
    pixel_seed = F(F(pixel_index) / F(F(W) * F(H)))
    indirect = np.zeros(3)
    bounces = []
    for i in range(gi_samples):
        throughput = ALBEDO.copy()
        ro_b, n_b = hit + 0.001 * n, n
        for b in range(gi_bounces):
            bounce_seed = F(F(F(b) * F(101.0)) + F(i))
            a17 = F(F(float(pixel_seed) * float(i) + 17.0) + TIME_H)     # fma(pixelSeed, f32(i), 17.0) + time: the product and sum are exact in f64
            a37 = F(F(float(pixel_seed) * float(i) + 37.0) + TIME_H)
            s1, s2 = rand2(bounce_seed, a17), rand2(a37, bounce_seed)
            d = sample_hemisphere(n_b, s1, s2)
            tb, nb, kb = closest(ro_b, d)
            rec = dict(i=i, b=b, s=[s1, s2], dir=d.tolist(), hit_tri=kb, t=tb if kb is not None else None)
            bounces.append(rec)
            if tb >= 1e19:
                break
            pb = ro_b + d * tb
            ds, occ_b, diff_b = direct(pb, nb, LIGHT_B, pixel_index, gi_samples, 4)
            rec.update(pos=pb.tolist(), direct=ds, occ=occ_b, diff=diff_b)
            indirect += throughput * ds
            throughput = throughput * ALBEDO
            ro_b, n_b = pb + 0.0001 * nb, nb
    indirect /= gi_samples
    log = []
    d0, occ0, diff0 = direct(hit, n, LIGHT_B, pixel_index, gi_samples, 4, log)
    col = ALBEDO * d0 + indirect
    return dict(pixel=PIXEL, pixel_index=pixel_index, hit_tri=tri, hit_t=t, bounces=bounces, indirect=indirect.tolist(),
                primary_samples=log, primary_occ=occ0, primary_diff=diff0, primary_direct=d0, col=col.tolist(), rgba8=rgba8(col))


if __name__ == "__main__":
    print(json.dumps(dict(case_a=case_a(), case_b=case_b()), indent=1))
