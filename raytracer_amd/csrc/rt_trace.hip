// rt_trace.hip — hot path 2: per-pixel ray/scene loop of web/compute.wgsl as a persistent-
// thread, wave64 traversal kernel for gfx950.
//
// A frame is four launches: k_cull (pixels whose primary ray misses the root box are final), k_scan + k_compact (the others
// go into a pixel list, hottest cost class first) and the persistent k_render: one lane = one list entry, each wave pulls
// 64-entry chunks from 64 sharded queue heads (home shard by XCD, then stealing) until the list - one frame's, this
// rank's stripes', or the lists of a whole batch of frames - is drained.  The traversal stack lives in LDS
// ([entry][thread], conflict-free) so that dynamic indexing never spills to scratch.
//
// Two node layouts (A/B, same results bit for bit):
//   WIDE64 (default)  64-byte records holding BOTH children's boxes + refs, triangles with
//                     hoisted edges: one 64 B fetch per internal step instead of the
//                     reference's 3 x 32 B (compute.wgsl:273,303-304), stack entries carry the
//                     child's entry distance so a pop needs no re-fetch.
//   AOS32             literal walk over the 32-byte interchange records (compute.wgsl:259-346).
// Semantics kept exactly: test order, near-child rule `b0.x <= b1.x`, the MAX_STACK-1 push
// guard that silently drops the far child (F9), strict `<` closest-hit update, front-face
// only Möller–Trumbore, child0-chain shadow walk (F8), shared RNG sequence (F7).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "rt_device.h"
#include "rt_internal.h"

namespace {

constexpr int TBLK = 256;
constexpr uint32_t REF_DONE = 0xFFFFFFFFu;
constexpr float TWO_PI_F = 6.283185307179586f;

struct Work { uint32_t box, tri, hit, drops; };
struct Hit { float t; uint32_t tri; float u, v; uint32_t steps; };   // steps: loop iterations of the traversal (scheduling hint only)

struct Scene {
  const float4* nodes;     // interchange records, 2 x float4 per node
  const float4* tri_pos;   // 3 x float4 per triangle
  const float4* tri_nor;
  const float4* wide;
  const float4* tri_pre;
  uint32_t root_ref;
  int leaf_shift;
  bool v2;                 // encoding of `nodes`
  int max_stack;
  uint32_t step_budget;    // > number of nodes: a valid tree never exhausts it; a cyclic (corrupt) import cannot hang the GPU
  const char* base; uint32_t wide_off, pre_off;   // RtRenderParams::near_*: 32-bit record offsets from one scalar base (NEAR kernels)
  uint2* stk;              // this thread's LDS column, stride TBLK: {reference, bits(entry distance)} - one 8-byte read per pop
  const float4* tri_uv;    // surface extension (rtcore.h "material / texture shading")
  const float4* materials;
  const RtTexDev* textures;
  uint32_t n_materials, n_textures;
};

// web/compute.wgsl:91-104
RT_DEV void slab(V3 ro, V3 inv, V3 mn, V3 mx, float& tmin, float& tmax) {
  const V3 t0 = (mn - ro) * inv, t1 = (mx - ro) * inv;
  tmin = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
  tmax = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
}
RT_DEV bool box_hit(float tmin, float tmax, float limit) { return (tmax >= fmaxf(tmin, 0.0f)) && (tmin <= limit); }

// web/compute.wgsl:107-132 with e1/e2 supplied (TRI = 0).  TRI = 1: the older shaders' test (web/shader.wgsl:133-163 =
// shader.wgsl:137-167) — the same quotients, other thresholds, and early returns whose outcome is this conjunction of
// the negated exit conditions (a NaN passes every exit and then fails `t < best` at the caller, as it does there).
template <int TRI>
RT_DEV bool moller_trumbore(V3 ro, V3 rd, V3 v0, V3 e1, V3 e2, float& t, float& u, float& v) {
  const V3 p = cross3(rd, e2);
  const float det = dot3(e1, p);
  const V3 s = ro - v0;
  const float invDet = 1.0f / det;
  u = dot3(s, p) * invDet;
  const V3 q = cross3(s, e1);
  v = dot3(rd, q) * invDet;
  t = dot3(e2, q) * invDet;
  if (TRI == 1) return !(det < 1e-8f) & !((u < 0.0f) | (u > 1.0f)) & !((v < 0.0f) | ((u + v) > 1.0f)) & !(t <= 1e-6f);
  return (det > 1e-10f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) & (t > 1e-10f);
}

// ---------------------------------------------------------------- WIDE64 traversal
// NEAR: both tables lie in one 4 GB window (RtRenderParams::near32): a record's address is scalar base + 32-bit byte offset, which
// the vector memory instructions take directly (global_load ... v_off, s[base]) - 8 full-rate address instructions per step instead
// of v_lshlrev_b64 / v_lshl_add_u64 / v_mad_u64_u32 behind exec-mask branches (-1.9 % of the shadow batch kernel).  A compile-time
// switch: with a run-time one the compiler merges the two forms' loads behind a pointer phi and both lose.
template <bool COUNT, int TRI = 0, bool NEAR = false>
RT_DEV Hit closest_wide(const Scene& S, V3 ro, V3 rd, V3 inv, float maxDist, Work& W) {
  float bestT = maxDist, bu = 0.0f, bv = 0.0f;
  uint32_t bestTri = 0xFFFFFFFFu;
  int sp = 0;
  uint32_t cur;
  {
    const float4 a = S.nodes[0], b = S.nodes[1];
    float tn, tf; slab(ro, inv, xyz(a), xyz(b), tn, tf);
    if (COUNT) W.box += 1;
    cur = box_hit(tn, tf, bestT) ? S.root_ref : REF_DONE;
  }
  auto pop = [&]() -> uint32_t {
    while (sp > 0) {
      sp -= 1;
      // read as ONE 64-bit word: the compiler splits a uint2 load into the distance (in the loop) and the reference (behind it),
      // two dependent LDS round trips per pop
      const unsigned long long e64 = reinterpret_cast<const unsigned long long*>(S.stk)[sp * TBLK];
      const uint2 e = make_uint2((uint32_t)e64, (uint32_t)(e64 >> 32));
      if (__uint_as_float(e.y) <= bestT) return e.x;  // re-test of compute.wgsl:275 with the stored entry distance
    }
    return REF_DONE;
  };
  // One loop, one memory round trip per iteration: a lane at an internal node fetches its 64-byte
  // record, a lane at a leaf fetches its triangle, from ONE set of loads.  (The descend-loop /
  // leaf-loop form made lanes that reached a leaf wait for the slowest descent and paid two
  // dependent round trips per visited leaf.)
  uint32_t budget = S.step_budget, leaf_j = 0;
  while (cur != REF_DONE) {
    if (--budget == 0) break;
    const bool is_leaf = (int)cur < 0;
    const uint32_t body = cur & 0x7FFFFFFFu;
    const uint32_t start = (S.leaf_shift >= 31) ? body : (body & ((1u << S.leaf_shift) - 1u));
    const uint32_t count = (S.leaf_shift >= 31) ? 1u : ((body >> S.leaf_shift) + 1u);
    const uint32_t ti = start + leaf_j;
    uint32_t off32 = 0;
    const float4* ptr;
    if constexpr (NEAR) {
      uint32_t ti3; asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(ti3) : "v"(ti));       // 3 * ti (48 * ti would become a v_mad_u64_u32)
      off32 = is_leaf ? (S.pre_off + (ti3 << 4)) : (S.wide_off + (cur << 6));
      ptr = reinterpret_cast<const float4*>(S.base + off32);
    } else ptr = is_leaf ? (S.tri_pre + 3 * (size_t)ti) : (S.wide + 4 * (size_t)cur);
    const float4 a = ptr[0], b = ptr[1], c = ptr[2];
    if (!is_leaf) {                                      // internal: both children from one 64 B record
      const float4 d = NEAR ? *reinterpret_cast<const float4*>(S.base + (off32 + 48u)) : ptr[3];
      float t0n, t0f, t1n, t1f;
      slab(ro, inv, xyz(a), xyz(b), t0n, t0f);
      slab(ro, inv, xyz(c), xyz(d), t1n, t1f);
      if (COUNT) W.box += 2;
      const bool hit0 = box_hit(t0n, t0f, bestT), hit1 = box_hit(t1n, t1f, bestT);
      const uint32_t r0 = __float_as_uint(a.w), r1 = __float_as_uint(b.w);
      if (hit0 && hit1) {
        const bool first0 = (t0n <= t1n);              // compute.wgsl:310
        if (sp < S.max_stack - 1) {                    // :311 / :317 — otherwise the far child is dropped
          S.stk[sp * TBLK] = make_uint2(first0 ? r1 : r0, __float_as_uint(first0 ? t1n : t0n));
          sp += 1;
        } else if (COUNT) W.drops += 1;
        cur = first0 ? r0 : r1;
      } else if (hit0) cur = r0;
      else if (hit1) cur = r1;
      else cur = pop();
    } else {
      if (COUNT) W.tri += 1;
      if (!(dot3(v3(a.w, b.w, c.w), rd) > 0.0f)) {      // back-face pre-cull, compute.wgsl:282
        float t, u, v;
        const bool hit = moller_trumbore<TRI>(ro, rd, xyz(a), xyz(b), xyz(c), t, u, v);
        if (hit && t < bestT) { bestT = t; bestTri = ti; bu = u; bv = v; if (COUNT) W.hit += 1; }
      }
      leaf_j += 1;
      if (leaf_j >= count) { leaf_j = 0; cur = pop(); }
    }
  }
  Hit h; h.t = bestT; h.tri = bestTri; h.u = bu; h.v = bv; h.steps = S.step_budget - budget;
  return h;
}

template <bool COUNT, int TRI = 0, bool NEAR = false>
RT_DEV bool occluded_wide(const Scene& S, V3 ro, V3 rd, float maxDist, Work& W) {
  const V3 inv = rcp3(rd);
  {
    const float4 a = S.nodes[0], b = S.nodes[1];
    float tn, tf; slab(ro, inv, xyz(a), xyz(b), tn, tf);
    if (COUNT) W.box += 1;
    if (!box_hit(tn, tf, maxDist)) return false;
  }
  uint32_t cur = S.root_ref;
  uint32_t budget = S.step_budget;
  while ((int)cur >= 0) {                              // child0 chain only (compute.wgsl:187)
    if (--budget == 0) return false;
    const float4* n = NEAR ? reinterpret_cast<const float4*>(S.base + (uint32_t)(S.wide_off + (cur << 6))) : S.wide + 4 * (size_t)cur;
    const float4 a = n[0], b = n[1];
    float tn, tf; slab(ro, inv, xyz(a), xyz(b), tn, tf);
    if (COUNT) W.box += 1;
    if (!box_hit(tn, tf, maxDist)) return false;
    cur = __float_as_uint(a.w);
  }
  const uint32_t body = cur & 0x7FFFFFFFu;
  const uint32_t start = (S.leaf_shift >= 31) ? body : (body & ((1u << S.leaf_shift) - 1u));
  const uint32_t count = (S.leaf_shift >= 31) ? 1u : ((body >> S.leaf_shift) + 1u);
  for (uint32_t j = 0; j < count; ++j) {
    const uint32_t ti = start + j;
    const float4 q0 = S.tri_pre[3 * (size_t)ti + 0], q1 = S.tri_pre[3 * (size_t)ti + 1], q2 = S.tri_pre[3 * (size_t)ti + 2];
    if (COUNT) W.tri += 1;
    float t, u, v;
    if (moller_trumbore<TRI>(ro, rd, xyz(q0), xyz(q1), xyz(q2), t, u, v) && t < maxDist) return true;
  }
  return false;
}

// ---------------------------------------------------------------- AOS32 traversal (literal)
struct NodeRec { V3 mn, mx; int c0, c1; bool leaf; int start, count; };
RT_DEV NodeRec fetch_node(const Scene& S, int idx) {   // compute.wgsl:68-88
  const float4 a = S.nodes[2 * (size_t)(uint32_t)idx], b = S.nodes[2 * (size_t)(uint32_t)idx + 1];
  NodeRec n;
  n.mn = xyz(a); n.mx = xyz(b);
  n.leaf = dec_is_leaf(S.v2, b.w);
  if (n.leaf) { n.start = dec_index(S.v2, a.w); n.count = dec_count(S.v2, b.w); n.c0 = -1; n.c1 = -1; }
  else { n.c0 = dec_index(S.v2, a.w); n.c1 = dec_index(S.v2, b.w); n.start = -1; n.count = 0; }
  return n;
}

template <bool COUNT, int TRI = 0>
RT_DEV Hit closest_aos(const Scene& S, V3 ro, V3 rd, V3 inv, float maxDist, Work& W) {
  float bestT = maxDist, bu = 0.0f, bv = 0.0f;
  uint32_t bestTri = 0xFFFFFFFFu;
  int sp = 0, cur = 0;
  if (COUNT) W.box += 1;
  uint32_t budget = 2u * S.step_budget;
  for (;;) {
    if (cur < 0 || sp >= S.max_stack) break;
    if (--budget == 0) break;
    const NodeRec node = fetch_node(S, cur);
    float hx, hy; slab(ro, inv, node.mn, node.mx, hx, hy);
    bool go_pop = false;
    if (box_hit(hx, hy, bestT)) {
      if (node.leaf) {
        for (int j = 0; j < node.count; ++j) {
          const uint32_t ti = (uint32_t)(node.start + j);
          const V3 va = xyz(S.tri_pos[3 * (size_t)ti + 0]), vb = xyz(S.tri_pos[3 * (size_t)ti + 1]), vc = xyz(S.tri_pos[3 * (size_t)ti + 2]);
          if (COUNT) W.tri += 1;
          const V3 e1 = vb - va, e2 = vc - va;
          if (dot3(cross3(e1, e2), rd) > 0.0f) continue;
          float t, u, v;
          const bool hit = moller_trumbore<TRI>(ro, rd, va, e1, e2, t, u, v);
          if (hit && t < bestT) { bestT = t; bestTri = ti; bu = u; bv = v; if (COUNT) W.hit += 1; }
        }
        go_pop = true;
      } else {
        const NodeRec n0 = fetch_node(S, node.c0), n1 = fetch_node(S, node.c1);
        float b0x, b0y, b1x, b1y;
        slab(ro, inv, n0.mn, n0.mx, b0x, b0y);
        slab(ro, inv, n1.mn, n1.mx, b1x, b1y);
        if (COUNT) W.box += 2;
        const bool hit0 = box_hit(b0x, b0y, bestT), hit1 = box_hit(b1x, b1y, bestT);
        if (hit0 && hit1) {
          const bool first0 = (b0x <= b1x);
          if (sp < S.max_stack - 1) { S.stk[sp * TBLK].x = (uint32_t)(first0 ? node.c1 : node.c0); sp += 1; }
          else if (COUNT) W.drops += 1;
          cur = first0 ? node.c0 : node.c1;
        } else if (hit0) cur = node.c0;
        else if (hit1) cur = node.c1;
        else go_pop = true;
      }
    } else go_pop = true;
    if (go_pop) {
      if (sp > 0) { sp -= 1; cur = (int)S.stk[sp * TBLK].x; } else break;
    }
  }
  Hit h; h.t = bestT; h.tri = bestTri; h.u = bu; h.v = bv; h.steps = 2u * S.step_budget - budget;
  return h;
}

template <bool COUNT, int TRI = 0>
RT_DEV bool occluded_aos(const Scene& S, V3 ro, V3 rd, float maxDist, Work& W) {   // compute.wgsl:166-194
  const V3 inv = rcp3(rd);
  int cur = 0;
  uint32_t budget = S.step_budget;
  for (;;) {
    if (cur < 0) break;
    if (--budget == 0) break;
    const NodeRec node = fetch_node(S, cur);
    float hx, hy; slab(ro, inv, node.mn, node.mx, hx, hy);
    if (COUNT) W.box += 1;
    if (!box_hit(hx, hy, maxDist)) break;
    if (node.leaf) {
      for (int j = 0; j < node.count; ++j) {
        const uint32_t ti = (uint32_t)(node.start + j);
        const V3 va = xyz(S.tri_pos[3 * (size_t)ti + 0]), vb = xyz(S.tri_pos[3 * (size_t)ti + 1]), vc = xyz(S.tri_pos[3 * (size_t)ti + 2]);
        if (COUNT) W.tri += 1;
        float t, u, v;
        if (moller_trumbore<TRI>(ro, rd, va, vb - va, vc - va, t, u, v) && t < maxDist) return true;
      }
      break;
    }
    cur = node.c0;
  }
  return false;
}

template <int LAYOUT, bool COUNT, int TRI = 0>
RT_DEV Hit closest(const Scene& S, V3 ro, V3 rd, V3 inv, float maxDist, Work& W) {
  if constexpr (LAYOUT != RT_LAYOUT_AOS32) return closest_wide<COUNT, TRI, LAYOUT == RT_LAYOUT_WIDE64_NEAR>(S, ro, rd, inv, maxDist, W);
  else return closest_aos<COUNT, TRI>(S, ro, rd, inv, maxDist, W);
}
template <int LAYOUT, bool COUNT, int TRI = 0>
RT_DEV bool occluded(const Scene& S, V3 ro, V3 rd, float maxDist, Work& W) {
  if constexpr (LAYOUT != RT_LAYOUT_AOS32) return occluded_wide<COUNT, TRI, LAYOUT == RT_LAYOUT_WIDE64_NEAR>(S, ro, rd, maxDist, W);
  else return occluded_aos<COUNT, TRI>(S, ro, rd, maxDist, W);
}

// interpolated shading normal of the winning triangle (compute.wgsl:288-291), evaluated once
RT_DEV V3 hit_normal(const Scene& S, const Hit& h) {
  const V3 n0 = xyz(S.tri_nor[3 * (size_t)h.tri + 0]), n1 = xyz(S.tri_nor[3 * (size_t)h.tri + 1]), n2 = xyz(S.tri_nor[3 * (size_t)h.tri + 2]);
  return normalize3(fma3(n2, h.v, fma3(n1, h.u, n0 * ((1.0f - h.u) - h.v))));
}

// the same normal as the older shaders write it, without fma (web/shader.wgsl:309, shader.wgsl:227)
RT_DEV V3 hit_normal_plain(const Scene& S, const Hit& h) {
  const V3 n0 = xyz(S.tri_nor[3 * (size_t)h.tri + 0]), n1 = xyz(S.tri_nor[3 * (size_t)h.tri + 1]), n2 = xyz(S.tri_nor[3 * (size_t)h.tri + 2]);
  return normalize3((n0 * ((1.0f - h.u) - h.v) + n1 * h.u) + n2 * h.v);
}

// Albedo of the surface at a hit: the definition in include/rtcore.h ("material / texture shading"), the same
// statements as the CPU checker restates them.
RT_DEV V3 hit_albedo(const Scene& S, const Hit& h) {
  const V3 ref = v3(1.0f, 0.84f, 0.0f);                                  // compute.wgsl:372
  const float4 a = S.tri_uv[2 * (size_t)h.tri + 0], b = S.tri_uv[2 * (size_t)h.tri + 1];
  const int mat = (int)__float_as_uint(b.z);
  if (mat < 0 || (uint32_t)mat >= S.n_materials) return ref;
  const float4 m = S.materials[mat];
  const V3 kd = xyz(m);
  const uint32_t slot = __float_as_uint(m.w);
  if (slot >= S.n_textures) return kd;
  const float w0 = (1.0f - h.u) - h.v;
  const float tu = __fmaf_rn(b.x, h.v, __fmaf_rn(a.z, h.u, a.x * w0));
  const float tv = __fmaf_rn(b.y, h.v, __fmaf_rn(a.w, h.u, a.y * w0));
  const float fu = tu - floorf(tu), fv = tv - floorf(tv);
  const RtTexDev t = S.textures[slot];
  const uint32_t x = min(sat_u32(fu * (float)t.w), t.w - 1u), y = min(sat_u32((1.0f - fv) * (float)t.h), t.h - 1u);
  const uint32_t px = t.texels[(size_t)y * t.w + x];
  return kd * v3((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f);
}

// ---------------------------------------------------------------- shading
struct Shade { int mode, gi_samples, gi_bounces, soft_samples; };
struct RayCount { uint32_t primary, shadow, bounce; };

// web/compute.wgsl:153-163
RT_DEV V3 sample_hemisphere(V3 n, float s1, float s2) {
  const float theta = TWO_PI_F * s1;
  const float r = sqrtf(s2);
  float sn, cs; rt_sincos(theta, &sn, &cs);
  const float x = r * cs, y = r * sn;
  const float z = sqrtf(fmaxf(0.0f, 1.0f - s2));
  const V3 up = (fabsf(n.y) < 0.999f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
  const V3 T = normalize3(cross3(up, n));
  const V3 B = cross3(n, T);
  return normalize3((T * x + B * y) + n * z);
}

// web/compute.wgsl:197-256
template <int LAYOUT, bool COUNT>
RT_DEV float direct_radiance(const Scene& S, const Shade& sh, V3 hit_os, V3 n_os, V3 light_os, uint32_t seed_id,
                             float time_h, Work& W, RayCount& R) {
  const V3 hitToLight = light_os - hit_os;
  const V3 L = normalize3(hitToLight);
  const float diff = fmaxf(dot3(n_os, L), 0.0f);
  if (diff <= 0.0f) return 0.0f;
  if (sh.mode == RT_MODE_PRIMARY) return diff;
  uint32_t occ = 0;
  const V3 up = (fabsf(n_os.y) > 0.99f) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f);
  const V3 tangent = normalize3(cross3(up, n_os));
  const V3 bitangent = cross3(n_os, tangent);
  const V3 ro2 = fma3(n_os, 0.001f, hit_os);
  const float lightDist = length3(hitToLight);
  const uint32_t GI = (uint32_t)sh.gi_samples, SOFT = (uint32_t)sh.soft_samples;
  const float sqrtS = ceilf(sqrtf((float)GI));
  const uint32_t cols = sat_u32(sqrtS);
  const uint32_t rows = (GI + cols - 1u) / cols;
  for (uint32_t i = 0; i < SOFT; ++i) {
    const uint32_t sx = i % cols, sy = i / cols;
    const float jitter1 = rand2((float)(seed_id + 17u) + time_h, (float)i + 0.37f);
    const float jitter2 = rand2((float)(seed_id + 91u) + time_h, (float)i + 1.13f);
    const float s1 = ((float)sx + jitter1) / (float)cols;
    const float s2 = ((float)sy + jitter2) / (float)rows;
    const float theta = TWO_PI_F * s1;
    const float r = sqrtf(s2);
    float sn, cs; rt_sincos(theta, &sn, &cs);
    const float x = r * cs, y = r * sn;
    const float z = sqrtf(fmaxf(0.0f, 1.0f - s2));
    const V3 rd2 = normalize3((tangent * x + bitangent * y) + n_os * z);
    const bool visible = dot3(rd2, L) > 0.5f;
    R.shadow += 1;
    const bool hit = occluded<LAYOUT, COUNT>(S, ro2, rd2, lightDist - 0.001f, W);
    occ += (visible & hit) ? 1u : 0u;
  }
  const float shadowFactor = __fmaf_rn(-(float)occ, 1.0f / (float)SOFT, 1.0f);
  return diff * shadowFactor;
}

RT_DEV V3 mat_mul4(const float* m, float x, float y, float z, float w) {
  return v3(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w, ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
            ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w);
}
RT_DEV V3 mat_mul3(const float* m, V3 v) {
  return v3((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z, (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
RT_DEV uint32_t unorm8(float c) { return sat_u32(floorf(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f + 0.5f)); }

struct PixelOut { uint32_t rgba; uint32_t tri; float t; uint32_t cost; };   // cost: traversal steps of all closest-hit rays of the pixel

struct PrimaryRay { V3 ro, rd, inv, light; uint32_t pixelIndex; float uvx, uvy; };

// web/compute.wgsl:419-442 (cs_main) + :351-361 (world -> object space)
// frag != 0: vs_main / generateRay / fs_main of the older shaders (web/shader.wgsl:61-67,166-173,435-440): the quad's uv at
// the fragment centre — framebuffer row 0 is the top row, clip-space y points up, so uv.y = 1 - (gy + 0.5) / h — and plain
// mul / add.  (A rasteriser's interpolation of uv is implementation-defined in its last bits; this is the definition.)
RT_DEV PrimaryRay primary_ray(const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h, int frag = 0) {
  const float uvx = ((float)gx + 0.5f) / (float)w, uvy = ((float)gy + 0.5f) / (float)h;
  const V3 ro_w = v3(U[0], U[1], U[2]);
  const float sx = __fmaf_rn(uvx, 2.0f, -1.0f), sy = __fmaf_rn(uvy, 2.0f, -1.0f);
  V3 rd_w = normalize3(fma3(v3(U[8], U[9], U[10]), sx, fma3(v3(U[12], U[13], U[14]), sy, v3(U[16], U[17], U[18]))));
  const float fuy = 1.0f - uvy;
  if (frag) {
    const float fsx = uvx * 2.0f - 1.0f, fsy = fuy * 2.0f - 1.0f;
    rd_w = normalize3((v3(U[8], U[9], U[10]) * fsx + v3(U[12], U[13], U[14]) * fsy) + v3(U[16], U[17], U[18]));
  }
  const float px = fminf(fmaxf(uvx * (float)w, 0.0f), (float)(w - 1));
  const float py = fminf(fmaxf(uvy * (float)h, 0.0f), (float)(h - 1));
  PrimaryRay r;
  r.uvx = uvx; r.uvy = fuy;
  r.pixelIndex = sat_u32(py) * w + sat_u32(px);
  const float* model = U + 20;
  r.ro = mat_mul4(model, ro_w.x, ro_w.y, ro_w.z, 1.0f);
  r.rd = normalize3(mat_mul3(model, rd_w));
  r.inv = rcp3(r.rd);
  r.light = mat_mul4(model, U[4], U[5], U[6], U[7]);
  return r;
}

// web/compute.wgsl:349-410
// GI = false: the same function compiled without the path-tracing loop, for launches that never run RT_MODE_PT (the loop's live
// state costs the batch kernel ~30 spilled registers under its 96-VGPR cap: 0.230 -> 0.218 ms per frame)
template <int LAYOUT, bool COUNT, bool TEX, bool GI = true>
RT_DEV PixelOut shade_pixel(const Scene& S, const Shade& sh, const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h,
                            float time_h, Work& W, RayCount& R) {
  const PrimaryRay pr = primary_ray(U, gx, gy, w, h);
  const V3 ro_os = pr.ro, rd_os = pr.rd, light_os = pr.light;
  const uint32_t pixelIndex = pr.pixelIndex;

  PixelOut out;
  R.primary += 1;
  const Hit hit0 = closest<LAYOUT, COUNT>(S, ro_os, rd_os, pr.inv, 1e20f, W);
  out.tri = hit0.tri; out.t = hit0.t; out.cost = hit0.steps;
  if (hit0.t >= 1e19f) { out.rgba = 0xFF000000u; out.tri = 0xFFFFFFFFu; return out; }   // (0,0,0,1)

  const V3 hit_pos = fma3(rd_os, hit0.t, ro_os);
  const V3 n_os = hit_normal(S, hit0);
  const V3 albedo = TEX ? hit_albedo(S, hit0) : v3(1.0f, 0.84f, 0.0f);
  V3 indirect = v3(0.0f, 0.0f, 0.0f);
  if (GI && sh.mode == RT_MODE_PT) {
    const uint32_t GIS = (uint32_t)sh.gi_samples, NB = (uint32_t)sh.gi_bounces;
    const float pixelSeed = (float)pixelIndex / (U[36] * U[37]);
    for (uint32_t i = 0; i < GIS; ++i) {
      V3 throughput = albedo;
      V3 ro_b = fma3(n_os, 0.001f, hit_pos);
      V3 n_b = n_os;
      for (uint32_t b = 0; b < NB; ++b) {
        const float bounceSeed = (float)b * 101.0f + (float)i;
        const float s1 = rand2(bounceSeed, __fmaf_rn(pixelSeed, (float)i, 17.0f) + time_h);
        const float s2 = rand2(__fmaf_rn(pixelSeed, (float)i, 37.0f) + time_h, bounceSeed);
        const V3 dir = sample_hemisphere(n_b, s1, s2);
        R.bounce += 1;
        const Hit hb = closest<LAYOUT, COUNT>(S, ro_b, dir, rcp3(dir), 1e20f, W);
        out.cost += hb.steps;
        if (hb.t >= 1e19f) break;
        const V3 pB = fma3(dir, hb.t, ro_b);
        const V3 nB = hit_normal(S, hb);
        const float dS = direct_radiance<LAYOUT, COUNT>(S, sh, pB, nB, light_os, pixelIndex, time_h, W, R);
        indirect = indirect + throughput * dS;
        throughput = throughput * (TEX ? hit_albedo(S, hb) : albedo);      // :400
        ro_b = fma3(nB, 0.0001f, pB);
        n_b = nB;
      }
    }
    indirect = indirect / (float)GIS;
  }
  const float d0 = direct_radiance<LAYOUT, COUNT>(S, sh, hit_pos, n_os, light_os, pixelIndex, time_h, W, R);
  const V3 col = v3(__fmaf_rn(albedo.x, d0, indirect.x), __fmaf_rn(albedo.y, d0, indirect.y), __fmaf_rn(albedo.z, d0, indirect.z));
  const float g = 1.0f / 2.2f;
  out.rgba = unorm8(rt_pow(col.x, g)) | (unorm8(rt_pow(col.y, g)) << 8) | (unorm8(rt_pow(col.z, g)) << 16) | 0xFF000000u;
  return out;
}

// The pixel's colour from its primary hit: compute.wgsl:369-409 without the GI loop (RT_MODE_PRIMARY / SHADOW), for the shade
// phase of the traverse / shade split.  The traversal phase left only the winning triangle's index; t, u, v come from running
// triIntersect_fast on that one triangle again - the same function on the same inputs (primary_ray is deterministic), hence
// the same bits the traversal held in bestT / u / v when it accepted the triangle.  That repeat is not a triangle TEST of the
// walk and is not counted.
template <int LAYOUT, bool COUNT, bool TEX>
RT_DEV uint32_t shade_known_hit(const Scene& S, const Shade& sh, const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h,
                                uint32_t tri, float time_h, Work& W, RayCount& R) {
  const PrimaryRay pr = primary_ray(U, gx, gy, w, h);
  Hit hit0; hit0.tri = tri; hit0.steps = 0;
  if constexpr (LAYOUT != RT_LAYOUT_AOS32) {
    const float4 q0 = S.tri_pre[3 * (size_t)tri + 0], q1 = S.tri_pre[3 * (size_t)tri + 1], q2 = S.tri_pre[3 * (size_t)tri + 2];
    (void)moller_trumbore<0>(pr.ro, pr.rd, xyz(q0), xyz(q1), xyz(q2), hit0.t, hit0.u, hit0.v);
  } else {
    const V3 va = xyz(S.tri_pos[3 * (size_t)tri + 0]), vb = xyz(S.tri_pos[3 * (size_t)tri + 1]), vc = xyz(S.tri_pos[3 * (size_t)tri + 2]);
    (void)moller_trumbore<0>(pr.ro, pr.rd, va, vb - va, vc - va, hit0.t, hit0.u, hit0.v);
  }
  const V3 hit_pos = fma3(pr.rd, hit0.t, pr.ro);
  const V3 n_os = hit_normal(S, hit0);
  const V3 albedo = TEX ? hit_albedo(S, hit0) : v3(1.0f, 0.84f, 0.0f);
  const float d0 = direct_radiance<LAYOUT, COUNT>(S, sh, hit_pos, n_os, pr.light, pr.pixelIndex, time_h, W, R);
  const V3 col = v3(__fmaf_rn(albedo.x, d0, 0.0f), __fmaf_rn(albedo.y, d0, 0.0f), __fmaf_rn(albedo.z, d0, 0.0f));   // indirect = (0,0,0), :408
  const float g = 1.0f / 2.2f;
  return unorm8(rt_pow(col.x, g)) | (unorm8(rt_pow(col.y, g)) << 8) | (unorm8(rt_pow(col.z, g)) << 16) | 0xFF000000u;
}

// ---- older fragment-pipeline shaders: web/shader.wgsl (V = 1) and the root shader.wgsl (V = 2) ------------
// Same traversal loop, child0-chain shadow walk and node / triangle buffers as compute.wgsl; they differ in the triangle
// test (moller_trumbore<1>), plain mul / add where compute.wgsl writes fma, the sky colour on a miss, an 8-tap Poisson-disk
// area light, their RNGs and the GI estimator.  Constant albedo (both files), WIDE64 records.
__constant__ float c_poisson[8][2] = {{-0.7071f, 0.7071f}, {0.7071f, 0.7071f}, {0.7071f, -0.7071f}, {-0.7071f, -0.7071f},
                                      {-0.3536f, 0.9239f}, {0.9239f, 0.3536f}, {0.3536f, -0.9239f}, {-0.9239f, -0.3536f}};   // web/shader.wgsl:27-36

// web/shader.wgsl:240-270 = shader.wgsl:355-384
template <bool COUNT>
RT_DEV float direct_radiance_disk(const Scene& S, const Shade& sh, V3 hit_os, V3 n_os, V3 light_os, float radius, Work& W, RayCount& R) {
  const V3 L = normalize3(light_os - hit_os);
  const float diff = fmaxf(dot3(n_os, L), 0.0f);
  if (diff <= 0.0f) return 0.0f;
  uint32_t occ = 0;
  const V3 perp1 = normalize3(cross3(n_os, L));
  const V3 perp2 = cross3(L, perp1);
  const uint32_t SOFT = (uint32_t)sh.soft_samples;
  const V3 ro2 = hit_os + n_os * 0.001f;
  for (uint32_t i = 0; i < SOFT; ++i) {
    const float ox = c_poisson[i][0] * radius, oy = c_poisson[i][1] * radius;
    const V3 samplePos = (light_os + perp1 * ox) + perp2 * oy;
    const V3 d = samplePos - hit_os;
    const V3 rd2 = normalize3(d);
    const float sd = length3(d);
    R.shadow += 1;
    if (occluded_wide<COUNT, 1>(S, ro2, rd2, sd - 0.001f, W)) occ += 1;
  }
  const float shadowFactor = 1.0f - (float)occ / (float)SOFT;
  return diff * shadowFactor;
}

RT_DEV bool is_safe_vec(V3 v) {   // shader.wgsl:170-172
  return (length3(v) < 1e19f) && (fabsf(v.x) <= 1e30f) && (fabsf(v.y) <= 1e30f) && (fabsf(v.z) <= 1e30f);
}

// shader.wgsl:45-50,334-352 : HEMI_SAMPLES[idx] = (((idx % 8) * 2 + 1) / 16, ((idx / 8) * 2 + 1) / 16), exact in f32
RT_DEV V3 sample_hemisphere_fixed(V3 n, uint32_t idx) {
  const float sx = (float)((idx & 7u) * 2u + 1u) / 16.0f, sy = (float)((idx >> 3) * 2u + 1u) / 16.0f;
  const float theta = sx * 6.28318530718f;
  const float phi = rt_acos(sqrtf(sy));
  float sinP, cosP, sinT, cosT;
  rt_sincos(phi, &sinP, &cosP);
  rt_sincos(theta, &sinT, &cosT);
  const V3 up = (fabsf(n.y) < 0.999f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
  const V3 T = normalize3(cross3(up, n));
  const V3 B = cross3(n, T);
  return normalize3((T * (sinP * cosT) + n * cosP) + B * (sinP * sinT));
}

// web/shader.wgsl:369-433 (V = 1) ; shader.wgsl:387-447 (V = 2)
template <bool COUNT, int V>
RT_DEV PixelOut shade_frag(const Scene& S, const Shade& sh, const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h,
                           uint32_t miss_rgba, Work& W, RayCount& R) {
  const PrimaryRay pr = primary_ray(U, gx, gy, w, h, V);
  const V3 light_os = pr.light;
  const float radius = U[7];
  PixelOut out;
  R.primary += 1;
  const Hit hit0 = closest_wide<COUNT, 1>(S, pr.ro, pr.rd, pr.inv, 1e20f, W);
  out.tri = hit0.tri; out.t = hit0.t; out.cost = hit0.steps;
  if (hit0.t >= 1e19f) { out.rgba = miss_rgba; out.tri = 0xFFFFFFFFu; return out; }       // sky
  const V3 hit_pos = pr.ro + pr.rd * hit0.t;
  V3 n_os = hit_normal_plain(S, hit0);
  const V3 albedo = v3(1.0f, 0.84f, 0.0f);
  const uint32_t GI = (uint32_t)sh.gi_samples;
  V3 col;
  if (V == 1) {
    const float directScalar = direct_radiance_disk<COUNT>(S, sh, hit_pos, n_os, light_os, radius, W, R);
    const V3 Lo = albedo * directScalar;
    V3 indirect = v3(0.0f, 0.0f, 0.0f);
    const uint32_t NB = (uint32_t)sh.gi_bounces;
    for (uint32_t i = 0; i < GI; ++i) {
      V3 throughput = albedo;
      V3 ro_b = hit_pos + n_os * 0.001f;
      V3 n_b = n_os;
      for (uint32_t b = 0; b < NB; ++b) {
        const float rx = frag_rand2<1>((float)i, (float)b + 17.0f);
        const float ry = frag_rand2<1>((float)i + 37.0f, (float)b);
        const V3 dir = sample_hemisphere(n_b, rx, ry);          // :189-204, the statements of compute.wgsl:153-163
        R.bounce += 1;
        const Hit h1 = closest_wide<COUNT, 1>(S, ro_b, dir, rcp3(dir), 50.0f, W);
        out.cost += h1.steps;
        if (h1.t >= 50.0f) break;
        const V3 p1 = ro_b + dir * h1.t;
        const V3 nn1 = hit_normal_plain(S, h1);
        const float dS = direct_radiance_disk<COUNT>(S, sh, p1, nn1, light_os, radius, W, R);
        indirect = indirect + throughput * dS;
        throughput = throughput * albedo;
        ro_b = p1 + nn1 * 0.001f;
        n_b = nn1;
      }
    }
    indirect = indirect / (float)GI;
    col = Lo + indirect;
  } else {
    if (!is_safe_vec(n_os)) n_os = v3(0.0f, 1.0f, 0.0f);
    const float direct = direct_radiance_disk<COUNT>(S, sh, hit_pos, n_os, light_os, radius, W, R);
    float indirect = 0.0f;
    const float giProb = 0.9f;
    for (uint32_t i = 0; i < GI; ++i) {
      const float sx = pr.uvx + (float)i * 0.031f, sy = pr.uvy + (float)i * 0.047f;
      if (frag_rand2<2>(sx, sy) > giProb) continue;
      const V3 dir1 = sample_hemisphere_fixed(n_os, i);
      const V3 ro1 = hit_pos + n_os * 0.001f;
      R.bounce += 1;
      const Hit h1 = closest_wide<COUNT, 1>(S, ro1, dir1, rcp3(dir1), 50.0f, W);
      out.cost += h1.steps;
      if (h1.t < 50.0f) {
        const V3 p1 = ro1 + dir1 * h1.t;
        V3 nn1 = hit_normal_plain(S, h1);
        if (!is_safe_vec(nn1)) nn1 = v3(0.0f, 1.0f, 0.0f);
        indirect = indirect + direct_radiance_disk<COUNT>(S, sh, p1, nn1, light_os, radius, W, R) / giProb;
      }
    }
    indirect = indirect / (float)GI;
    const float k = direct + indirect * 0.6f;
    col = albedo * k;
  }
  const float g = 1.0f / 2.2f;
  out.rgba = unorm8(rt_pow(col.x, g)) | (unorm8(rt_pow(col.y, g)) << 8) | (unorm8(rt_pow(col.z, g)) << 16) | 0xFF000000u;
  return out;
}

// ---- PT split: the pixel's work of compute.wgsl:349-410 in two parts with identical arithmetic -----------------------
// A PT pixel is one chain of GI_SAMPLES x GI_BOUNCES dependent traversals (cfg 4: 48 closest-hit + up to 240 shadow walks,
// ~4 ms) while the whole frame is ~2.5 ms of work: with one lane per pixel the launch was 5.2 ms, half of it a tail of
// few waves.  The GI samples of a pixel are independent chains; only their SUM has an order (`indirect +=` runs over i,
// then b, compute.wgsl:376-404).  Part 1 = everything outside the GI loop; part 2 = one lane per (pixel, sample), then the
// terms are added in exactly that order by part 3.
struct PtRec { V3 hit_pos, n_os, albedo; float d0; uint32_t idx, pixelIndex; };

template <int LAYOUT, bool COUNT, bool TEX>
RT_DEV PixelOut shade_pt_primary(const Scene& S, const Shade& sh, const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h,
                                 float time_h, Work& W, RayCount& R, PtRec& rec, bool& is_hit) {
  const PrimaryRay pr = primary_ray(U, gx, gy, w, h);
  PixelOut out;
  R.primary += 1;
  const Hit hit0 = closest<LAYOUT, COUNT>(S, pr.ro, pr.rd, pr.inv, 1e20f, W);
  out.tri = hit0.tri; out.t = hit0.t; out.cost = hit0.steps; out.rgba = 0xFF000000u;
  is_hit = !(hit0.t >= 1e19f);
  if (!is_hit) { out.tri = 0xFFFFFFFFu; return out; }                    // (0,0,0,1), compute.wgsl:365-367
  rec.hit_pos = fma3(pr.rd, hit0.t, pr.ro);
  rec.n_os = hit_normal(S, hit0);
  rec.albedo = TEX ? hit_albedo(S, hit0) : v3(1.0f, 0.84f, 0.0f);
  rec.pixelIndex = pr.pixelIndex;
  rec.d0 = direct_radiance<LAYOUT, COUNT>(S, sh, rec.hit_pos, rec.n_os, pr.light, pr.pixelIndex, time_h, W, R);   // :408
  return out;
}

// One GI sample i of a pixel: compute.wgsl:378-403.  term[b] = throughput * dS of bounce b, n_terms of them.
template <int LAYOUT, bool COUNT, bool TEX>
RT_DEV void pt_sample_chain(const Scene& S, const Shade& sh, const float* U, const PtRec& rec, V3 light_os, uint32_t i, float time_h,
                            Work& W, RayCount& R, V3 term[4], uint32_t& n_terms) {
  const uint32_t NB = (uint32_t)sh.gi_bounces;
  const float pixelSeed = (float)rec.pixelIndex / (U[36] * U[37]);
  V3 throughput = rec.albedo;
  V3 ro_b = fma3(rec.n_os, 0.001f, rec.hit_pos);
  V3 n_b = rec.n_os;
  n_terms = 0;
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) {
    if (b >= NB) break;
    const float bounceSeed = (float)b * 101.0f + (float)i;
    const float s1 = rand2(bounceSeed, __fmaf_rn(pixelSeed, (float)i, 17.0f) + time_h);
    const float s2 = rand2(__fmaf_rn(pixelSeed, (float)i, 37.0f) + time_h, bounceSeed);
    const V3 dir = sample_hemisphere(n_b, s1, s2);
    R.bounce += 1;
    const Hit hb = closest<LAYOUT, COUNT>(S, ro_b, dir, rcp3(dir), 1e20f, W);
    if (hb.t >= 1e19f) break;
    const V3 pB = fma3(dir, hb.t, ro_b);
    const V3 nB = hit_normal(S, hb);
    const float dS = direct_radiance<LAYOUT, COUNT>(S, sh, pB, nB, light_os, rec.pixelIndex, time_h, W, R);
    term[b] = throughput * dS;
    n_terms = b + 1u;
    throughput = throughput * (TEX ? hit_albedo(S, hb) : rec.albedo);      // :400
    ro_b = fma3(nB, 0.0001f, pB);
    n_b = nB;
  }
}

RT_DEV uint32_t wave_sum_u32(uint32_t v) {
  for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}

RT_DEV uint32_t local_to_global_row(const RtRenderParams& p, uint32_t lrow) {
  if (p.stripe_rows <= 0) return lrow;
  const uint32_t sl = lrow / (uint32_t)p.stripe_rows, rin = lrow % (uint32_t)p.stripe_rows;
  return (sl * (uint32_t)p.stripe_world + (uint32_t)p.stripe_rank) * (uint32_t)p.stripe_rows + rin;
}

// Per-block partial counters, plain stores into the block's own slot (a same-address atomic per
// wave costs ~12 ns each and serialises: 8192 waves x 2 counters were ~0.2 ms of a 0.36 ms launch).
template <bool COUNT>
RT_DEV void flush_stats(const RtRenderParams& p, int lane, const RayCount& R, const Work& W, uint32_t pixels) {
  __shared__ uint32_t s_part[TBLK / 64][8];
  uint32_t c[8] = {R.primary, R.shadow, R.bounce, W.box, W.tri, W.hit, pixels, W.drops};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t s = (!COUNT && (k == 3 || k == 4 || k == 5 || k == 7)) ? 0u : wave_sum_u32(c[k]);
    if (lane == 0) s_part[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned long long t = 0;
    for (int w = 0; w < TBLK / 64; ++w) t += s_part[w][threadIdx.x];
    p.stats_partial[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
  }
}

// A batch with RT_FLAG_COUNT_WORK: a wave's counters belong to the frame of the chunk it is working on; when the next chunk is
// another frame's (and at the end) the wave adds them to ITS slot of that frame, stats_partial[frame][wave of the grid][8] -
// a slot has one writer, so plain read-modify-write, no atomics - and starts from zero again.
RT_DEV void flush_batch_stats(const RtRenderParams& p, uint32_t frame, int lane, RayCount& R, Work& W, uint32_t& pixels) {
  const uint32_t c[8] = {R.primary, R.shadow, R.bounce, W.box, W.tri, W.hit, pixels, W.drops};
  const size_t slot = ((size_t)frame * (gridDim.x * (TBLK / 64)) + blockIdx.x * (TBLK / 64) + (threadIdx.x >> 6)) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t s = wave_sum_u32(c[k]);
    if (lane == 0 && s) p.stats_partial[slot + k] += s;
  }
  R.primary = R.shadow = R.bounce = 0; W.box = W.tri = W.hit = W.drops = 0; pixels = 0;
}
// grid.x = frame: stats[frame][k] = sum over the frame's wave slots + the pixels k_cull disposed of (as k_stats_final)
__global__ void k_stats_final_batch(const RtRenderParams p, const RtBatch bt, uint32_t n_slots, uint32_t n_slots_b, uint32_t n_inside, int count_work) {
  __shared__ unsigned long long s_sum[8][32];
  const uint32_t f = blockIdx.x;
  const int k = threadIdx.x & 7, part = threadIdx.x >> 3;
  unsigned long long t = 0;
  for (uint32_t b = part; b < n_slots; b += 32) t += p.stats_partial[((size_t)f * n_slots + b) * 8 + k];
  const unsigned long long* pb = p.stats_partial + (size_t)8 * gridDim.x * n_slots;          // the shade phase's region (traverse / shade split)
  for (uint32_t b = part; b < n_slots_b; b += 32) t += pb[((size_t)f * n_slots_b + b) * 8 + k];
  s_sum[k][part] = t;
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned long long tot = 0;
    for (int i = 0; i < 32; ++i) tot += s_sum[threadIdx.x][i];
    const uint32_t listed = p.list_count[(size_t)bt.frames[f].src * bt.lc_stride];
    const unsigned long long culled = (unsigned long long)n_inside - (unsigned long long)listed;
    if (threadIdx.x == 0 || threadIdx.x == 6 || (threadIdx.x == 3 && count_work)) tot += culled;
    p.stats[(size_t)f * 9 + threadIdx.x] = tot;
    if (threadIdx.x == 0) p.stats[(size_t)f * 9 + 8] = listed;       // pixels_traced
  }
}

// stats[k] = sum of the phase-B block partials; pixels culled in phase A never reach phase B:
// each of them is one primary ray, one root-box test, one pixel.
__global__ void k_stats_final(const RtRenderParams p, uint32_t n_blocks, uint32_t n_inside, int count_work) {
  __shared__ unsigned long long s_sum[8][32];
  const int k = threadIdx.x & 7, part = threadIdx.x >> 3;     // 256 threads: 8 counters x 32 strided readers
  unsigned long long t = 0;
  for (uint32_t b = part; b < n_blocks; b += 32) t += p.stats_partial[(size_t)b * 8 + k];
  s_sum[k][part] = t;
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned long long tot = 0;
    for (int i = 0; i < 32; ++i) tot += s_sum[threadIdx.x][i];
    const unsigned long long culled = (unsigned long long)n_inside - (unsigned long long)(*p.list_count);
    if (threadIdx.x == 0 || threadIdx.x == 6 || (threadIdx.x == 3 && count_work)) tot += culled;
    p.stats[threadIdx.x] = tot;
  }
}

// ---- frame pipeline: k_cull -> k_compact -> k_render (persistent) -> k_stats_final -------------
// Same-address atomics serialise at ~26 ns each on this part (measured: 10 k list appends cost
// 267 us of a 325 us launch), so the compaction is atomic-free and the phase-B queue is sharded.
//
// Frame-coherent scheduling.  A 64-pixel chunk costs what its slowest lane costs, and pixel costs are spread over two
// orders of magnitude (cfg 3: a ray that only crosses the root box takes ~6 traversal steps, a hit ~40, a grazing ray up to
// 185).  With the pixel list in plain tile order the queue was empty at 361 us of a 523 us launch: the rest was a tail of
// few long chunks, and silhouette tiles kept 60 cheap lanes idle behind 4 long ones.  The previous frame measured every
// pixel (k_render stores a cost class per pixel); the next frame's k_cull / k_compact build the list as
// [hot | normal | cheap]: long rays share waves, start first, and the tail of the launch is made of cheap chunks.  The
// classes are hints only: they decide the ORDER pixels are traced in, never a result, and a frame without hints
// (first frame, resized image) is simply traced in tile order.
constexpr int CULL_TPB = 64;          // tiles per k_cull / k_compact block: 16 rounds x 4 waves
constexpr int N_SHARD = 64;           // phase-B queue shards, one 128-byte line of queue state each
constexpr int SHARD_STRIDE = 32;      // uint32 words between shard heads: one 128-byte line each
constexpr int N_CLASS = RT_N_CLASS;   // cost classes of the pixel list: 0 cheap / unknown, 1 normal, 2 .. N-1 quantile bands of the non-cheap pixels
static_assert(N_CLASS >= 3 && N_CLASS <= 8, "class_share[] holds N_CLASS - 2 <= 6 entries");
static int shard_stride_words() { return SHARD_STRIDE; }


// Phase A1 — cull.  A primary ray that misses the root box (compute.wgsl:275 on node 0) can never hit anything: its pixel
// is final here (the miss colour, compute.wgsl:365-367).  Two levels, because one lane per pixel with a full ray set-up
// (two normalisations, nine divisions) made this kernel 35 us at cfg 3, where nearly every ray enters the box:
//   level 1, lane = tile: the rays through the tile's 4 corner pixels.  Rays from one origin that hit a convex box form a
//     convex cone and a pixel's (un-normalised) direction is a convex combination of its tile's corner directions, so a
//     tile whose 4 corner rays enter needs no per-pixel test.  (Calling a pixel "entering" is always safe: k_render tests
//     the root box itself and writes the same miss.  Only "culled" must be certain, and that stays a per-pixel f32 test.)
//   level 2, lane = one 8-pixel ROW of a tile (a wave covers 8 tiles): the ray test only in silhouette tiles; last
//     frame's 8 cost classes of the row are one 8-byte load.  The row's 8-bit masks (entering pixels by class) are bytes
//     of the tile's 64-bit masks in LDS — bit 8 * row + x, the order k_compact and k_render expect.
// What differs between the frames of a batch (rt_render_batch_device); a single frame's view is the RtRenderParams itself.
struct FrameView {
  const float* U; uint8_t* rgba8; uint32_t* hit_tri; float* hit_t; int cull_x0, cull_x1, cull_y0, cull_y1;
  unsigned long long* masks; uint32_t* block_total; uint32_t* list; uint32_t* list_count;
  uint32_t* hits;          // traverse / shade split: per list entry the primary hit's triangle (0xFFFFFFFF = miss), this frame's own region
  bool dup;                // a batch frame whose 44 uniforms equal an earlier frame's: it shares that frame's masks / list (k_cull still writes ITS culled pixels)
};
template <bool BATCH, typename B>
RT_DEV FrameView frame_view(const RtRenderParams& p, const B& bt, uint32_t f) {
  FrameView v;
  if constexpr (BATCH) {
    const RtFrameRec& r = bt.frames[f];                    // f is wave-uniform: scalar loads
    v.U = r.U; v.rgba8 = r.rgba8; v.hit_tri = r.hit_tri; v.hit_t = r.hit_t;
    v.cull_x0 = r.cull[0]; v.cull_x1 = r.cull[1]; v.cull_y0 = r.cull[2]; v.cull_y1 = r.cull[3];
    const uint32_t g = r.src;                              // the frame whose cull / compaction result this one uses (itself unless a duplicate view)
    v.masks = p.masks + (size_t)g * bt.mask_stride; v.block_total = p.block_total + (size_t)g * bt.bt_stride;
    v.list = p.list + (size_t)g * bt.list_stride; v.list_count = p.list_count + (size_t)g * bt.lc_stride;
    v.hits = p.hits ? p.hits + (size_t)f * bt.list_stride : nullptr;
    v.dup = g != f;
  } else {
    v.dup = false; v.hits = p.hits;
    v.U = p.U; v.rgba8 = p.rgba8; v.hit_tri = p.hit_tri; v.hit_t = p.hit_t;
    v.cull_x0 = p.cull_x0; v.cull_x1 = p.cull_x1; v.cull_y0 = p.cull_y0; v.cull_y1 = p.cull_y1;
    v.masks = p.masks; v.block_total = p.block_total; v.list = p.list; v.list_count = p.list_count;
  }
  return v;
}

RT_DEV void cull_write_miss(const RtRenderParams& p, const FrameView& fv, size_t idx0, uint32_t bits) {
  const bool vec = bits == 0xFFu && (p.width & 3u) == 0u;      // a whole row of a whole tile: 2 x 16-byte stores per output
  if (fv.rgba8) {
    uint32_t* d = reinterpret_cast<uint32_t*>(fv.rgba8) + idx0;
    if (vec && (reinterpret_cast<uintptr_t>(fv.rgba8) & 15u) == 0u) {
      const uint4 v = make_uint4(p.miss_rgba, p.miss_rgba, p.miss_rgba, p.miss_rgba);
      reinterpret_cast<uint4*>(d)[0] = v; reinterpret_cast<uint4*>(d)[1] = v;
    } else for (int i = 0; i < 8; ++i) if ((bits >> i) & 1u) d[i] = p.miss_rgba;
  }
  if (fv.hit_tri) {
    uint32_t* d = fv.hit_tri + idx0;
    if (vec && (reinterpret_cast<uintptr_t>(fv.hit_tri) & 15u) == 0u) {
      const uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      reinterpret_cast<uint4*>(d)[0] = v; reinterpret_cast<uint4*>(d)[1] = v;
    } else for (int i = 0; i < 8; ++i) if ((bits >> i) & 1u) d[i] = 0xFFFFFFFFu;
  }
  if (fv.hit_t) {
    float* d = fv.hit_t + idx0;
    if (vec && (reinterpret_cast<uintptr_t>(fv.hit_t) & 15u) == 0u) {
      const float4 v = make_float4(1e20f, 1e20f, 1e20f, 1e20f);
      reinterpret_cast<float4*>(d)[0] = v; reinterpret_cast<float4*>(d)[1] = v;
    } else for (int i = 0; i < 8; ++i) if ((bits >> i) & 1u) d[i] = 1e20f;
  }
}

template <bool BATCH, typename B>
__global__ __launch_bounds__(TBLK) void k_cull(const RtRenderParams p, const B bt) {
  const FrameView fv = frame_view<BATCH>(p, bt, BATCH ? blockIdx.y : 0u);
  __shared__ uint8_t s_corner[4][CULL_TPB];
  __shared__ uint32_t s_txy[CULL_TPB];
  __shared__ unsigned long long s_m[N_CLASS][CULL_TPB];      // the block's tile masks; byte r of a mask comes from the lane of tile row r
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4 ra = reinterpret_cast<const float4*>(p.nodes)[0], rb = reinterpret_cast<const float4*>(p.nodes)[1];
  if (blockIdx.x == 0 && blockIdx.y == 0)   // phase-B queue heads start at zero (k_render runs two launches later)
    for (uint32_t i = threadIdx.x; i < (uint32_t)(N_SHARD * SHARD_STRIDE); i += TBLK) {
      p.shard_heads[i] = 0u;
      if (p.shade_heads) p.shade_heads[i] = 0u;
      if (p.pt_cap) { p.pt_heads[i] = 0u; if (i < (uint32_t)N_SHARD) p.pt_count[i] = 0u; }
    }
  bool any_in = false;
  {   // level 1: wave w traces corner w of the block's 64 tiles (0: x0 y0, 1: x7 y0, 2: x0 y7, 3: x7 y7)
    const uint32_t tile = blockIdx.x * CULL_TPB + (uint32_t)lane;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;   // once per tile: <= 8192 tiles per axis (resolution <= 65536)
    if (w == 0) s_txy[lane] = tx | (ty << 16);
    uint8_t state = 0;                                      // 0 = outside the root box's screen rectangle (or no tile), 1 = corner ray enters, 2 = corner ray misses
    if (tile < p.n_tiles) {
      const int tx0 = (int)(tx * 8u), ty0 = (int)local_to_global_row(p, ty * 8u);   // a tile's 8 rows are contiguous in the image: stripes are multiples of 8 rows
      const bool tile_in = tx0 <= fv.cull_x1 && tx0 + 7 >= fv.cull_x0 && ty0 <= fv.cull_y1 && ty0 + 7 >= fv.cull_y0;
      if (tile_in) {
        const PrimaryRay pr = primary_ray(fv.U, (uint32_t)tx0 + ((w & 1) ? 7u : 0u), (uint32_t)ty0 + ((w & 2) ? 7u : 0u), p.width, p.height, p.frag);
        float tn, tf; slab(pr.ro, pr.inv, xyz(ra), xyz(rb), tn, tf);
        state = box_hit(tn, tf, 1e20f) ? 1 : 2;
      }
    }
    s_corner[w][lane] = state;
    any_in = state != 0;
  }
  if (!__syncthreads_or(any_in ? 1 : 0)) {
    // Every tile of the block lies outside the root box's screen rectangle (4 of 5 blocks of cfg 3's frame): nothing to test and
    // nothing to list - the block only writes the miss value, and does so row-major: the 16-byte pieces of one image row of its 64
    // tiles in consecutive lanes, so that a wave's store instruction is ONE contiguous 1 KB run (lane = tile row makes it eight
    // runs of 256 bytes; contiguous runs store at 5.6 TB/s against 4.4, tools/ubench_stream).
    const bool vec = (p.width & 3u) == 0u;
#pragma unroll 1
    for (uint32_t i = 0; i < 4u; ++i) {
      const uint32_t pc = i * TBLK + threadIdx.x;              // piece: 4 pixels
      const uint32_t row = pc >> 7, k = (pc >> 1) & 63u, half = pc & 1u;
      const uint32_t tile = blockIdx.x * CULL_TPB + k;
      if (tile >= p.n_tiles) continue;
      const uint32_t txy = s_txy[k], x0 = (txy & 0xFFFFu) * 8u + half * 4u, lrow = (txy >> 16) * 8u + row;
      const uint32_t y = local_to_global_row(p, lrow);
      if (!(lrow < p.local_rows && y < p.height) || x0 >= p.width) continue;
      const uint32_t nv = min(4u, p.width - x0);
      const size_t idx0 = (size_t)lrow * p.width + x0;
      if (fv.rgba8) {
        uint32_t* d = reinterpret_cast<uint32_t*>(fv.rgba8) + idx0;
        if (vec && nv == 4u && (reinterpret_cast<uintptr_t>(fv.rgba8) & 15u) == 0u) *reinterpret_cast<uint4*>(d) = make_uint4(p.miss_rgba, p.miss_rgba, p.miss_rgba, p.miss_rgba);
        else for (uint32_t j = 0; j < nv; ++j) d[j] = p.miss_rgba;
      }
      if (fv.hit_tri) {
        uint32_t* d = fv.hit_tri + idx0;
        if (vec && nv == 4u && (reinterpret_cast<uintptr_t>(fv.hit_tri) & 15u) == 0u) *reinterpret_cast<uint4*>(d) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        else for (uint32_t j = 0; j < nv; ++j) d[j] = 0xFFFFFFFFu;
      }
      if (fv.hit_t) {
        float* d = fv.hit_t + idx0;
        if (vec && nv == 4u && (reinterpret_cast<uintptr_t>(fv.hit_t) & 15u) == 0u) *reinterpret_cast<float4*>(d) = make_float4(1e20f, 1e20f, 1e20f, 1e20f);
        else for (uint32_t j = 0; j < nv; ++j) d[j] = 1e20f;
      }
    }
    // empty totals; the tiles' masks are not written: k_compact sees from the totals that the block lists nothing and never reads them
    if (!fv.dup && threadIdx.x < N_CLASS) fv.block_total[(size_t)blockIdx.x * N_CLASS + threadIdx.x] = 0u;
    return;
  }
  uint8_t* sm8 = reinterpret_cast<uint8_t*>(&s_m[0][0]);
  const int row = lane & 7;
#pragma unroll 1
  for (int r = 0; r < CULL_TPB / (8 * (TBLK / 64)); ++r) {
    const int k = (r * (TBLK / 64) + w) * 8 + (lane >> 3);     // tile of the block this lane's row belongs to
    const uint32_t tile = blockIdx.x * CULL_TPB + (uint32_t)k;
    uint32_t e = 0, cm[N_CLASS];                                // 8-bit row masks: entering pixels; of them, last frame's class c (cm[0] is filled last)
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) cm[c] = 0u;
    if (tile < p.n_tiles) {
      const uint32_t txy = s_txy[k], x0 = (txy & 0xFFFFu) * 8u, lrow = (txy >> 16) * 8u + (uint32_t)row;
      const uint32_t y = local_to_global_row(p, lrow);
      uint32_t vm = 0;                                          // pixels of the row that exist
      if (lrow < p.local_rows && y < p.height) vm = (p.width - x0 >= 8u) ? 0xFFu : ((1u << (p.width - x0)) - 1u);
      const uint32_t cs = (uint32_t)s_corner[0][k] | ((uint32_t)s_corner[1][k] << 8) | ((uint32_t)s_corner[2][k] << 16) | ((uint32_t)s_corner[3][k] << 24);
      if (cs == 0x01010101u) e = vm;
      else if ((cs & 0xFFu) != 0u && vm) {                      // silhouette tile: the reference's own test, per pixel
#pragma unroll 1
        for (uint32_t i = 0; i < 8u; ++i) {
          if (!((vm >> i) & 1u)) continue;
          const PrimaryRay pr = primary_ray(fv.U, x0 + i, y, p.width, p.height, p.frag);
          float tn, tf; slab(pr.ro, pr.inv, xyz(ra), xyz(rb), tn, tf);
          if (box_hit(tn, tf, 1e20f)) e |= 1u << i;
        }
      }
      const size_t idx0 = (size_t)lrow * p.width + x0;
      if (vm & ~e) cull_write_miss(p, fv, idx0, vm & ~e);            // ray / box-test / pixel counts of these: k_stats_final
      if (e && p.pix_class) {                                    // last frame's hints; anything else than 1 / 2 counts as 0
        unsigned long long cls = 0ull;
        if ((p.width & 7u) == 0u) cls = *reinterpret_cast<const unsigned long long*>(p.pix_class + idx0);
        else for (uint32_t i = 0; i < 8u; ++i) if ((vm >> i) & 1u) cls |= (unsigned long long)p.pix_class[idx0 + i] << (8u * i);
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
          const uint32_t b = (uint32_t)(cls >> (8u * i)) & 0xFFu;
#pragma unroll
          for (int c = 1; c < N_CLASS; ++c) cm[c] |= (b == (uint32_t)c ? 1u : 0u) << i;
        }
#pragma unroll
        for (int c = 1; c < N_CLASS; ++c) cm[c] &= e;
      }
    }
    if (p.class_promote) {                                       // the hottest pixel of a tile gives its class to the tile's non-cheap pixels
      unsigned long long has[N_CLASS];
#pragma unroll
      for (int c = 2; c < N_CLASS; ++c) has[c] = __ballot(cm[c] != 0u);
      int top = 0;
#pragma unroll
      for (int c = 2; c < N_CLASS; ++c) if ((has[c] >> (lane & 56)) & 0xFFull) top = c;
      if (top) {
        uint32_t all = 0u;
#pragma unroll
        for (int c = 1; c < N_CLASS; ++c) { all |= cm[c]; cm[c] = 0u; }
#pragma unroll
        for (int c = 2; c < N_CLASS; ++c) if (c == top) cm[c] = all;
      }
    }
    uint32_t any = 0u;
#pragma unroll
    for (int c = 1; c < N_CLASS; ++c) any |= cm[c];
    cm[0] = e & ~any;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) sm8[((size_t)(c * CULL_TPB + k)) * 8 + row] = (uint8_t)cm[c];
  }
  __syncthreads();
  for (int c = w; c < N_CLASS; c += TBLK / 64) {                 // lane = tile: masks out, per-class totals of the block
    const uint32_t tile = blockIdx.x * CULL_TPB + (uint32_t)lane;
    const unsigned long long m = s_m[c][lane];
    if (fv.dup) continue;                                        // the earlier frame of this view writes the (identical) masks and totals
    if (tile < p.n_tiles) fv.masks[(size_t)tile * N_CLASS + c] = m;
    const uint32_t tot = wave_sum_u32((uint32_t)__popcll(m));
    if (lane == 0) fv.block_total[(size_t)blockIdx.x * N_CLASS + c] = tot;
  }
}

// Phase A1b — scan.  k_cull left one total per block and cost class; one workgroup per frame turns them, in place, into
// exclusive prefixes over the blocks and writes the frame's list counts: [0] entries, [1] entries of the hottest class,
// [8 + c] / [16 + c] start / length of class c's run (list order: hottest class first, cheap / unknown (0) last).
// (k_compact used to fold ALL block totals in EVERY block: 2025 blocks x 32 KB = 65 MB of L2 reads per 4K frame.)
template <bool BATCH, typename B>
__global__ __launch_bounds__(TBLK) void k_scan(const RtRenderParams p, const B bt) {
  const FrameView fv = frame_view<BATCH>(p, bt, BATCH ? blockIdx.x : 0u);
  if (fv.dup) return;                                            // (block-uniform) one scan per distinct view
  __shared__ uint32_t s_wave[TBLK / 64][N_CLASS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t n_blocks = (p.n_tiles + CULL_TPB - 1) / CULL_TPB;      // = rt_cull_blocks(n_tiles), the grid of k_cull / k_compact
  const uint32_t per = (n_blocks + TBLK - 1) / TBLK;
  const uint32_t b0 = min((uint32_t)threadIdx.x * per, n_blocks), b1 = min(b0 + per, n_blocks);   // my consecutive blocks
  uint32_t sum[N_CLASS];
#pragma unroll
  for (int c = 0; c < N_CLASS; ++c) sum[c] = 0u;
  for (uint32_t b = b0; b < b1; ++b)
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) sum[c] += fv.block_total[(size_t)b * N_CLASS + c];
  uint32_t run[N_CLASS], tot[N_CLASS];
#pragma unroll
  for (int c = 0; c < N_CLASS; ++c) {                     // exclusive scan over the 256 threads: wave scan, then the waves
    uint32_t inc = sum[c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[w][c] = inc;
    run[c] = inc - sum[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < N_CLASS; ++c) {
    uint32_t before = 0u, all = 0u;
#pragma unroll
    for (int i = 0; i < TBLK / 64; ++i) { const uint32_t t = s_wave[i][c]; if (i < w) before += t; all += t; }
    run[c] += before; tot[c] = all;
  }
  for (uint32_t b = b0; b < b1; ++b)
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
      const uint32_t t = fv.block_total[(size_t)b * N_CLASS + c];
      fv.block_total[(size_t)b * N_CLASS + c] = run[c];
      run[c] += t;
    }
  if (threadIdx.x == 0) {
    uint32_t above = 0u;
#pragma unroll
    for (int c = N_CLASS - 1; c >= 0; --c) { fv.list_count[8 + c] = above; fv.list_count[16 + c] = tot[c]; above += tot[c]; }
    fv.list_count[0] = above; fv.list_count[1] = tot[N_CLASS - 1];
  }
}

// Phase A2 — compact.  Block b owns the same 64 tiles as in k_cull.  List layout: class-major (hot, normal, cheap); inside
// a class block-major, tile-major, lane order.  Every offset is a sum of k_cull's per-block per-class totals (a few KB
// read per block) plus a wave scan of the block's 64 tile popcounts: atomic-free and deterministic.
template <bool BATCH, typename B>
__global__ __launch_bounds__(TBLK) void k_compact(const RtRenderParams p, const B bt) {
  const FrameView fv = frame_view<BATCH>(p, bt, BATCH ? blockIdx.y : 0u);
  if (fv.dup) return;                                            // (block-uniform) one pixel list per distinct view
  __shared__ uint32_t s_pref[N_CLASS][CULL_TPB];
  __shared__ uint32_t s_txy[CULL_TPB];
  __shared__ unsigned long long s_mask[N_CLASS][CULL_TPB];   // the block's masks, read from HBM once (the rounds below took one dependent load each)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t me = blockIdx.x;
  constexpr int CPW = (N_CLASS + TBLK / 64 - 1) / (TBLK / 64);     // classes per wave: wave w scans classes w, w + 4
  unsigned long long my_mask[CPW];
  // A block that lists nothing (most of a frame: k_cull's blocks outside the root box's rectangle, which do not even write masks):
  // k_scan's exclusive prefixes say so - this block's share of class c is the next block's prefix minus its own.
  bool empty = true;
#pragma unroll
  for (int c = 0; c < N_CLASS; ++c) {
    const uint32_t a = fv.block_total[(size_t)me * N_CLASS + c];
    const uint32_t b = (me + 1u < gridDim.x) ? fv.block_total[(size_t)(me + 1u) * N_CLASS + c] : fv.list_count[16 + c];
    empty = empty && a == b;
  }
  if (empty && me != 0u) return;                                 // (block-uniform; block 0 still derives the frame's thresholds below)
  {
    const uint32_t tile = me * CULL_TPB + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = w + i * (TBLK / 64);
      my_mask[i] = (!empty && c < N_CLASS && tile < p.n_tiles) ? fv.masks[(size_t)tile * N_CLASS + c] : 0ull;
    }
    if (w == TBLK / 64 - 1) s_txy[lane] = (tile % p.tiles_x) | ((tile / p.tiles_x) << 16);   // tile coordinates, once per tile
  }
#pragma unroll
  for (int i = 0; i < CPW; ++i) {                             // a wave scans the block's 64 tile counts of its classes
    const int cl = w + i * (TBLK / 64);
    if (cl >= N_CLASS) break;
    const uint32_t c = (uint32_t)__popcll(my_mask[i]);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
      if (lane >= o) inc += t;
    }
    s_pref[cl][lane] = inc - c;
    s_mask[cl][lane] = my_mask[i];
  }
  __syncthreads();
  // list order: hottest class first, cheap / unknown (0) last; class starts and the prefixes over the blocks are k_scan's
  uint32_t base[N_CLASS];
#pragma unroll
  for (int c = 0; c < N_CLASS; ++c) base[c] = fv.list_count[8 + c] + fv.block_total[(size_t)me * N_CLASS + c];
  // (a batch shares one set of thresholds and one step histogram: frame 0's block derives them)
  if (me == 0 && w == TBLK / 64 - 1 && p.pix_class && (!BATCH || blockIdx.y == 0)) {   // hot threshold of THIS frame from last frame's step histogram
    // thresholds of classes 2 .. N-1 (list_count[2 ..]): fixed t2 -> one hot class; adaptive -> quantile bands
    uint32_t thr[N_CLASS];
#pragma unroll
    for (int c = 2; c < N_CLASS; ++c) thr[c] = (c == 2) ? p.class_t2 : 0xFFFFFFFFu;
    if (p.class_t2 == 0u) {                               // lane b owns bin b: steps in [t1 + 4 b, t1 + 4 b + 4), last bin open
      const unsigned long long mine = p.class_hist[lane];
      p.class_hist[lane] = 0u;                            // this frame's k_render (next launch) accumulates afresh
      unsigned long long suffix = mine;                   // pixels in bins >= lane
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_down(suffix, o);
        if (lane + o < 64) suffix += t;
      }
      const unsigned long long total = __shfl(suffix, 0);
#pragma unroll
      for (int c = 2; c < N_CLASS; ++c) {                 // first bin whose suffix fits into the class's share; no data -> no such class
        const unsigned long long ok = __ballot(total > 0ull && suffix * 100ull <= total * p.class_share[c - 2]);
        thr[c] = ok ? p.class_t1 + 4u * (uint32_t)(__ffsll((long long)ok) - 1) : 0xFFFFFFFFu;
      }
    }
    if (lane == 0) {
      uint32_t floor_t = p.class_t1 + 1u;                 // thresholds ascend with the class
#pragma unroll
      for (int c = 2; c < N_CLASS; ++c) { thr[c] = max(thr[c], floor_t); floor_t = thr[c]; fv.list_count[c] = thr[c]; }
    }
  }
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // most blocks of a frame lie outside the root box's screen rectangle and have nothing to list (cfg 3: 4 of 5)
  unsigned long long mine_any = 0ull;
#pragma unroll
  for (int i = 0; i < CPW; ++i) mine_any |= my_mask[i];
  if (!__syncthreads_or(mine_any != 0ull)) return;
#pragma unroll 1
  for (int r = 0; r < CULL_TPB / (TBLK / 64); ++r) {
    const int k = r * (TBLK / 64) + w;
    const uint32_t tile = me * CULL_TPB + (uint32_t)k;
    if (tile >= p.n_tiles) break;
    const uint32_t txy = s_txy[k], tx = txy & 0xFFFFu, ty = txy >> 16;
    const uint32_t x = tx * 8u + (uint32_t)(lane & 7), lrow = ty * 8u + (uint32_t)(lane >> 3);
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
      const unsigned long long m = s_mask[c][k];
      if ((m >> lane) & 1ull) fv.list[base[c] + s_pref[c][k] + (uint32_t)__popcll(m & lt)] = lrow * p.width + x;
    }
  }
}

// Phase B — persistent-thread traversal + shading over the compacted pixel list (all 64 lanes
// carry rays that enter the tree).  The list's 64-pixel chunks are dealt to N_SHARD queue shards in groups of p.shard_group
// consecutive chunks, round-robin, so every shard walks the list front to back (hot -> cheap) and all shards hold the
// same amount of work; each shard has its own head on its own cache line.  A wave drains its home shard
// (blocks with equal blockIdx % 8 share an XCD and get neighbouring shards = neighbouring chunk groups,
// for L2 locality only), then steals from whichever shard a cheap load scan shows non-empty.
RT_DEV uint32_t shard_chunk(uint32_t shard, uint32_t k, uint32_t group) {     // k-th chunk of a shard; increasing in k
  return ((k / group) * N_SHARD + shard) * group + (k % group);
}
#ifdef RT_TIMELINE   // analysis build only (make EXTRA=-DRT_TIMELINE): per-wave start / last-chunk-start / end clocks of k_render
__device__ unsigned long long g_timeline[4096 * 4][4];
extern "C" int rt_debug_timeline(unsigned long long* out, size_t n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * n_words);
}
#endif

// what a lane does with its list entry.  TRAVERSE + SHADE are the two launches RT_MODE_PRIMARY / SHADOW frames run as: the
// primary ray's walk alone (69 - 82 VGPRs: 5 - 6 waves per SIMD instead of the 3 of the whole pixel, and the launch's time follows
// the number of resident waves: profiles/r3/ab_occupancy_lite.txt), then the colour of the pixels that hit something
// Register caps (second __launch_bounds__ argument = waves per SIMD the compiler must make room for).  The launches of a batch
// have no tail, their time follows the number of resident waves, and a few spilled registers cost less than a
// missing wave: the whole-pixel kernel of a batch at <= 96 VGPRs (5 waves; 26 - 61 spilled) runs PT-16 batches in 2.54 ms per
// frame instead of 3.31 and shadow batches in 0.255 instead of 0.314 (profiles/r3/ab_register_cap.txt; 4 waves: 2.79 / 0.269,
// 6 waves: 2.59 / 0.273).  One frame per launch is as long as its longest chain whatever the occupancy: no cap there.
#ifndef RT_PIXEL_BATCH_MIN_WAVES
#define RT_PIXEL_BATCH_MIN_WAVES 5
#endif
#ifndef RT_PIXEL_NOGI_BATCH_MIN_WAVES
#define RT_PIXEL_NOGI_BATCH_MIN_WAVES 5
#endif
#ifndef RT_SHADE_BATCH_MIN_WAVES
#define RT_SHADE_BATCH_MIN_WAVES 5
#endif
#ifndef RT_PT_SAMPLES_MIN_WAVES
#define RT_PT_SAMPLES_MIN_WAVES 5                  // both launches of the per-frame PT split (KIND_PTSPLIT sizes the grid of both): PT-16 3.54 -> 3.34 ms per frame
#endif
enum { KIND_PIXEL = 0, KIND_PTSPLIT = 1, KIND_FRAG_WEB = 2, KIND_FRAG_ROOT = 3, KIND_TRAVERSE = 4, KIND_SHADE = 5,
       KIND_PIXEL_NOGI = 6 /* KIND_PIXEL without the path-tracing loop: RT_MODE_PRIMARY / SHADOW batches */ };
template <int LAYOUT, bool COUNT, bool TEX, int KIND, bool BATCH = false, typename B = RtNoBatch>
__global__ __launch_bounds__(TBLK, (BATCH && KIND == 6) ? RT_PIXEL_NOGI_BATCH_MIN_WAVES : (BATCH && KIND == 0) ? RT_PIXEL_BATCH_MIN_WAVES : (LAYOUT == RT_LAYOUT_AOS32) ? 1 : (BATCH && KIND == 5) ? RT_SHADE_BATCH_MIN_WAVES : (KIND == 1) ? RT_PT_SAMPLES_MIN_WAVES : 1) void k_render(const RtRenderParams p, const B bt) {
  constexpr bool PTSPLIT = KIND == KIND_PTSPLIT;
  constexpr bool SHADE = KIND == KIND_SHADE;
  static_assert(!BATCH || KIND != KIND_PTSPLIT, "a batch runs PT one lane per pixel (no PT split)");
  constexpr int RPL = (RT_MAX_BATCH * N_CLASS + 63) / 64;   // runs (frame x cost class) of a batch's run table per lane
  constexpr int N_RUN = RPL * 64;
  // A work item of the queue: 64 list entries - one per lane - for every kind but SHADE.  The shade launch takes SHADE_G x 64
  // consecutive entries of one run, keeps those whose primary ray hit something (about a quarter of cfg 3's list) in a
  // per-wave LDS queue and colours them 64 at a time: lanes = hit pixels, not list entries.
  constexpr uint32_t SHADE_G = 8;
  constexpr uint32_t ITEM = SHADE ? 64u * SHADE_G : 64u;
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef RT_TIMELINE
  const unsigned long long tl_start = wall_clock64();
  unsigned long long tl_last = tl_start, tl_chunks = 0, tl_max = 0;   // tl_max: the wave's longest chunk (10 ns ticks)
#endif
  uint2* stk = reinterpret_cast<uint2*>(smem);

  Scene S;
  S.nodes = reinterpret_cast<const float4*>(p.nodes);
  S.tri_pos = reinterpret_cast<const float4*>(p.tri_pos);
  S.tri_nor = reinterpret_cast<const float4*>(p.tri_nor);
  S.wide = p.wide; S.tri_pre = p.tri_pre;
  S.base = p.near_base; S.wide_off = p.near_wide_off; S.pre_off = p.near_pre_off;
  S.root_ref = p.root_ref; S.leaf_shift = p.leaf_shift; S.v2 = p.nodes_v2 != 0; S.max_stack = p.max_stack; S.step_budget = p.step_budget;
  S.stk = stk + threadIdx.x;
  S.tri_uv = p.tri_uv; S.materials = p.materials; S.textures = p.textures; S.n_materials = p.n_materials; S.n_textures = p.n_textures;
  Shade sh; sh.mode = p.mode; sh.gi_samples = p.gi_samples; sh.gi_bounces = p.gi_bounces; sh.soft_samples = p.soft_samples;

  const int lane = threadIdx.x & 63;
  float time_h = (float)hash_u32(sat_u32(p.U[38]));          // compute.wgsl:444 (a batch: per frame, below)
  Work W; W.box = W.tri = W.hit = W.drops = 0;
  RayCount R; R.primary = R.shadow = R.bounce = 0;
  uint32_t pixels = 0;
  uint32_t cur_frame = 0xFFFFFFFFu;                      // BATCH && COUNT: the frame the counters above belong to
  uint32_t class_thr[N_CLASS];                           // class_thr[c], c >= 2: steps from which a pixel is of class c (k_compact)
#pragma unroll
  for (int c = 2; c < N_CLASS; ++c) class_thr[c] = p.pix_class ? p.list_count[c] : 0xFFFFFFFFu;
  __shared__ uint32_t s_hist[64];                        // this block's step histogram of non-cheap pixels (next frame's threshold)
  if (threadIdx.x < 64) s_hist[threadIdx.x] = 0u;
  // The chunk space.  One frame: the list's 64-pixel chunks, front to back.  A batch: the runs (class, frame) of all the
  // frames' lists, hottest class of EVERY frame first (class-major, frame-minor), each run cut into its own chunks.  The
  // run table (first chunk, list start, entries, frame of every run) lives in LDS, one copy per block, built by wave 0
  // (until round 3 every lane held two runs in eight registers for the whole launch: registers are this kernel's occupancy);
  // a chunk id finds its run with one ballot per 64 runs over the runs' first chunks.
  uint32_t n_list = 0, n_chunks = 0;
  const uint32_t *seg_first_p = nullptr, *seg_start_p = nullptr, *seg_count_p = nullptr, *seg_frame_p = nullptr;
  if constexpr (!BATCH) {
    n_list = *p.list_count;                              // written by k_compact (previous launch on this stream)
    n_chunks = (n_list + ITEM - 1u) / ITEM;
    __syncthreads();
  } else {
    __shared__ uint32_t s_first[N_RUN], s_start[N_RUN], s_count[N_RUN], s_frame[N_RUN], s_total;
    if (threadIdx.x < 64) {
      const uint32_t n_seg = bt.n * (uint32_t)N_CLASS;
      uint32_t total = 0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const uint32_t run = (uint32_t)r * 64u + (uint32_t)lane;
        uint32_t my_chunks = 0, frame = 0, start = 0, count = 0;
        if (run < n_seg) {
          frame = run % bt.n;
          const uint32_t c = (uint32_t)(N_CLASS - 1) - run / bt.n;                        // hottest class first, cheap / unknown (0) last
          const uint32_t* lc = p.list_count + (size_t)bt.frames[frame].src * bt.lc_stride;
          start = lc[8 + c]; count = lc[16 + c];
          my_chunks = (count + ITEM - 1u) / ITEM;
        }
        uint32_t inc = my_chunks;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
        s_first[run] = (run < n_seg && my_chunks > 0u) ? total + inc - my_chunks : 0xFFFFFFFFu;   // empty runs never match
        s_start[run] = start; s_count[run] = count; s_frame[run] = frame;
        total += (uint32_t)__shfl((int)inc, 63);
      }
      if (lane == 0) s_total = total;
    }
    __syncthreads();
    n_chunks = s_total;
    // (closures below read the table through these)
    seg_first_p = s_first; seg_start_p = s_start; seg_count_p = s_count; seg_frame_p = s_frame;
  }
  uint32_t* const heads = SHADE ? p.shade_heads : p.shard_heads;
  const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
  int shard = (int)((blockIdx.x & 7u) * (N_SHARD / 8) + ((blockIdx.x >> 3) * (N_SHARD / 8)) / per_xcd) & (N_SHARD - 1);
  for (uint32_t scans = 0; scans <= 4u * N_SHARD; ++scans) {          // every wave reaches the exit: a scan that finds nothing
    // lane l looks after shard l
    const uint32_t head = __hip_atomic_load(heads + (size_t)lane * p.shard_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long avail = __ballot(shard_chunk((uint32_t)lane, head, p.shard_group) < n_chunks);
    if (!avail) break;
    // first non-empty shard at or after `shard` (rotating)
    const unsigned long long rot = (shard == 0) ? avail : ((avail >> shard) | (avail << (64 - shard)));
    shard = (shard + (__ffsll((long long)rot) - 1)) & (N_SHARD - 1);
    for (;;) {
      uint32_t h = 0;
      if (lane == 0) h = atomicAdd(heads + (size_t)shard * p.shard_stride, 1u);
      const uint32_t chunk = shard_chunk((uint32_t)shard, (uint32_t)__builtin_amdgcn_readfirstlane((int)h), p.shard_group);
      if (chunk >= n_chunks) break;
#ifdef RT_TIMELINE
      { const unsigned long long now = wall_clock64(); if (tl_chunks && now - tl_last > tl_max) tl_max = now - tl_last; tl_last = now; tl_chunks += 1; }
#endif
      // this chunk's frame and its entries of that frame's list
      FrameView fv = frame_view<false>(p, bt, 0u);
      uint32_t li, li_end;
      bool last_frame = true;
      uint32_t batch_frame = 0;
      if constexpr (!BATCH) { li = chunk * ITEM + (uint32_t)lane; li_end = n_list; }
      else {
        // the last run that starts at or before this chunk (empty runs never match); chunk is wave-uniform, so are the branches
        int o = 0;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const unsigned long long m = __ballot(seg_first_p[r * 64 + lane] <= chunk);
          if (m) o = r * 64 + 63 - __clzll((long long)m);
        }
        o = __builtin_amdgcn_readfirstlane(o);
        const uint32_t f = seg_frame_p[o], first = seg_first_p[o], start = seg_start_p[o], cnt = seg_count_p[o];
        fv = frame_view<true>(p, bt, f);
        li = start + (chunk - first) * ITEM + (uint32_t)lane; li_end = start + cnt;
        time_h = (float)hash_u32(sat_u32(fv.U[38]));
        last_frame = f + 1u == bt.n;
        batch_frame = f;
      }
      if constexpr (BATCH && COUNT) {                       // counters are per frame: hand over what the previous chunk's frame collected
        if (cur_frame != 0xFFFFFFFFu && cur_frame != batch_frame) flush_batch_stats(p, cur_frame, lane, R, W, pixels);
        cur_frame = batch_frame;
      }
      if constexpr (SHADE) {
        __shared__ uint32_t s_queue[TBLK / 64][2 * 64 * SHADE_G];          // per wave: (list entry, triangle) of the item's hits
        uint32_t* q = s_queue[threadIdx.x >> 6];
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        uint32_t tri_j[SHADE_G];
#pragma unroll
        for (uint32_t j = 0; j < SHADE_G; ++j) { const uint32_t e = li + 64u * j; tri_j[j] = e < li_end ? fv.hits[e] : 0xFFFFFFFFu; }
        uint32_t n_hit = 0;
#pragma unroll
        for (uint32_t j = 0; j < SHADE_G; ++j) {
          const bool is_hit = tri_j[j] != 0xFFFFFFFFu;
          const unsigned long long m = __ballot(is_hit);
          if (is_hit) { const uint32_t pos = n_hit + (uint32_t)__popcll(m & lt); q[2 * pos] = li + 64u * j; q[2 * pos + 1] = tri_j[j]; }
          n_hit += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t k0 = 0; k0 < n_hit; k0 += 64u) {
          const uint32_t k = k0 + (uint32_t)lane;
          if (k < n_hit) {
            const uint32_t e = q[2 * k], tri = q[2 * k + 1];
            const uint32_t idx = fv.list[e];
            const uint32_t x = idx % p.width, y = local_to_global_row(p, idx / p.width);
            const uint32_t rgba = shade_known_hit<LAYOUT, COUNT, TEX>(S, sh, fv.U, x, y, p.width, p.height, tri, time_h, W, R);
            if (fv.rgba8) reinterpret_cast<uint32_t*>(fv.rgba8)[idx] = rgba;
          }
        }
        __builtin_amdgcn_wave_barrier();                              // the queue is rewritten by the wave's next item
        continue;
      }
      PtRec rec;
      bool rec_hit = false;
      if (li < li_end) {
        const uint32_t idx = fv.list[li];
        const uint32_t x = idx % p.width, lrow = idx / p.width;
        const uint32_t y = local_to_global_row(p, lrow);
        PixelOut o;
        if constexpr (PTSPLIT) { o = shade_pt_primary<LAYOUT, COUNT, TEX>(S, sh, fv.U, x, y, p.width, p.height, time_h, W, R, rec, rec_hit); rec.idx = idx; }
        else if constexpr (KIND == KIND_TRAVERSE) {
          const PrimaryRay pr = primary_ray(fv.U, x, y, p.width, p.height);
          R.primary += 1;
          const Hit h0 = closest<LAYOUT, COUNT>(S, pr.ro, pr.rd, pr.inv, 1e20f, W);
          const bool miss = h0.t >= 1e19f;
          o.rgba = 0xFF000000u; o.tri = miss ? 0xFFFFFFFFu : h0.tri; o.t = h0.t; o.cost = h0.steps;   // a miss is final: (0,0,0,1), :365-367
          fv.hits[li] = o.tri;                                  // the shade phase colours the hits
          rec_hit = !miss;
        }

        else if constexpr (KIND == KIND_FRAG_WEB) o = shade_frag<COUNT, 1>(S, sh, fv.U, x, y, p.width, p.height, p.miss_rgba, W, R);
        else if constexpr (KIND == KIND_FRAG_ROOT) o = shade_frag<COUNT, 2>(S, sh, fv.U, x, y, p.width, p.height, p.miss_rgba, W, R);
        else if constexpr (KIND == KIND_PIXEL_NOGI) o = shade_pixel<LAYOUT, COUNT, TEX, false>(S, sh, fv.U, x, y, p.width, p.height, time_h, W, R);
        else o = shade_pixel<LAYOUT, COUNT, TEX>(S, sh, fv.U, x, y, p.width, p.height, time_h, W, R);
        {
        if (fv.rgba8 && !rec_hit) reinterpret_cast<uint32_t*>(fv.rgba8)[idx] = o.rgba;     // a hit pixel gets its colour in part 2 (PT split, traverse / shade split)
        if (fv.hit_tri) fv.hit_tri[idx] = o.tri;
        if (fv.hit_t) fv.hit_t[idx] = o.t;
        if (p.pix_class && last_frame) {                  // a batch leaves the hints of its last frame
          uint32_t cls = o.cost >= p.class_t1 ? 1u : 0u;
#pragma unroll
          for (int c = 2; c < N_CLASS; ++c) cls += o.cost >= class_thr[c] ? 1u : 0u;
          p.pix_class[idx] = (uint8_t)cls;
          if (o.cost >= p.class_t1) atomicAdd(&s_hist[min((o.cost - p.class_t1) >> 2, 63u)], 1u);
        }
        pixels += 1;
        }
      }
      if constexpr (PTSPLIT) {                              // append the chunk's hit records to the sub-list of its shard: one atomic per chunk
        const unsigned long long hm = __ballot(rec_hit);
        if (hm) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(p.pt_count + shard, (uint32_t)__popcll(hm));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (rec_hit) {
            const uint32_t r = base + (uint32_t)__popcll(hm & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            float4* d = p.pt_recs + ((size_t)shard * p.pt_cap + r) * 3;
            d[0] = make_float4(rec.hit_pos.x, rec.hit_pos.y, rec.hit_pos.z, rec.d0);
            d[1] = make_float4(rec.n_os.x, rec.n_os.y, rec.n_os.z, __uint_as_float(rec.idx));
            d[2] = make_float4(rec.albedo.x, rec.albedo.y, rec.albedo.z, __uint_as_float(rec.pixelIndex));
          }
        }
      }
    }
  }
#ifdef RT_TIMELINE
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long* t = g_timeline[blockIdx.x * 4 + (threadIdx.x >> 6)];
    const unsigned long long tl_end = wall_clock64();
    if (tl_chunks && tl_end - tl_last > tl_max) tl_max = tl_end - tl_last;
    t[0] = tl_start; t[1] = tl_last; t[2] = tl_end; t[3] = tl_chunks | (tl_max << 32);
  }
#endif
  if constexpr (!BATCH) flush_stats<COUNT>(p, lane, R, W, pixels);
  else if constexpr (COUNT) { if (cur_frame != 0xFFFFFFFFu) flush_batch_stats(p, cur_frame, lane, R, W, pixels); }   // a batch keeps statistics only when asked to count
  if (!SHADE && p.pix_class) {                           // 64 bins, one atomic per non-empty bin and block, spread over the launch's tail
    __syncthreads();
    if (threadIdx.x < 64 && s_hist[threadIdx.x]) atomicAdd(&p.class_hist[threadIdx.x], s_hist[threadIdx.x]);
  }
}

// PT part 2 — persistent.  A wave item = (64 records of one sub-list, ONE GI sample i): lane = pixel, as in k_render, so the
// 64 rays of a wave-step keep the coherence of neighbouring pixels that share the sample's random numbers (F7 / §8a-17) —
// giving the 64 lanes to 16 samples of 4 pixels instead was 25 % SLOWER than no split at all (incoherent gathers).
// The bounce terms throughput * dS go to HBM (SoA, coalesced); part 3 adds them in the reference's order.
template <int LAYOUT, bool COUNT, bool TEX>
__global__ __launch_bounds__(TBLK, LAYOUT == RT_LAYOUT_WIDE64 ? RT_PT_SAMPLES_MIN_WAVES : 1) void k_pt_samples(const RtRenderParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint2* stk = reinterpret_cast<uint2*>(smem);
  Scene S;
  S.nodes = reinterpret_cast<const float4*>(p.nodes);
  S.tri_pos = reinterpret_cast<const float4*>(p.tri_pos);
  S.tri_nor = reinterpret_cast<const float4*>(p.tri_nor);
  S.wide = p.wide; S.tri_pre = p.tri_pre;
  S.base = p.near_base; S.wide_off = p.near_wide_off; S.pre_off = p.near_pre_off;
  S.root_ref = p.root_ref; S.leaf_shift = p.leaf_shift; S.v2 = p.nodes_v2 != 0; S.max_stack = p.max_stack; S.step_budget = p.step_budget;
  S.stk = stk + threadIdx.x;
  S.tri_uv = p.tri_uv; S.materials = p.materials; S.textures = p.textures; S.n_materials = p.n_materials; S.n_textures = p.n_textures;
  Shade sh; sh.mode = p.mode; sh.gi_samples = p.gi_samples; sh.gi_bounces = p.gi_bounces; sh.soft_samples = p.soft_samples;

  const int lane = threadIdx.x & 63;
  const float time_h = (float)hash_u32(sat_u32(p.U[38]));          // compute.wgsl:444
  const V3 light_os = mat_mul4(p.U + 20, p.U[4], p.U[5], p.U[6], p.U[7]);   // :360, as primary_ray
  Work W; W.box = W.tri = W.hit = W.drops = 0;
  RayCount R; R.primary = R.shadow = R.bounce = 0;
  const uint32_t GI = (uint32_t)p.gi_samples, NB = (uint32_t)p.gi_bounces;
  const size_t cap_total = (size_t)N_SHARD * p.pt_cap;

  // lane s looks after sub-list s: ceil(count / 64) record chunks x GI samples, chunk-major
  const uint32_t my_cnt = p.pt_count[lane], my_items = ((my_cnt + 63u) / 64u) * GI;
  const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
  int shard = (int)((blockIdx.x & 7u) * (N_SHARD / 8) + ((blockIdx.x >> 3) * (N_SHARD / 8)) / per_xcd) & (N_SHARD - 1);
  for (uint32_t scans = 0; scans <= 4u * N_SHARD; ++scans) {          // every wave reaches the exit: a scan that finds nothing
    const uint32_t head = __hip_atomic_load(p.pt_heads + (size_t)lane * p.shard_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long avail = __ballot(head < my_items);
    if (!avail) break;
    const unsigned long long rot = (shard == 0) ? avail : ((avail >> shard) | (avail << (64 - shard)));
    shard = (shard + (__ffsll((long long)rot) - 1)) & (N_SHARD - 1);
    const uint32_t n_items = (uint32_t)__shfl((int)my_items, shard), cnt = (uint32_t)__shfl((int)my_cnt, shard);
    for (;;) {
      uint32_t h = 0;
      if (lane == 0) h = atomicAdd(p.pt_heads + (size_t)shard * p.shard_stride, 1u);
      const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
      if (item >= n_items) break;
      const uint32_t r = (item / GI) * 64u + (uint32_t)lane, i = item % GI;
      if (r < cnt) {
        const size_t g = (size_t)shard * p.pt_cap + r;
        const float4* d = p.pt_recs + g * 3;
        const float4 a = d[0], b4 = d[1], c = d[2];
        PtRec rec;
        rec.hit_pos = xyz(a); rec.d0 = a.w; rec.n_os = xyz(b4); rec.idx = __float_as_uint(b4.w);
        rec.albedo = xyz(c); rec.pixelIndex = __float_as_uint(c.w);
        V3 term[4];
        term[0] = term[1] = term[2] = term[3] = v3(0.0f, 0.0f, 0.0f);
        uint32_t n_terms = 0;
        pt_sample_chain<LAYOUT, COUNT, TEX>(S, sh, p.U, rec, light_os, i, time_h, W, R, term, n_terms);
        float* t = p.pt_terms + (size_t)i * (NB * 3u + 1u) * cap_total + g;     // [i][b][xyz | n][record]
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b)
          if (b < n_terms) { t[(b * 3u + 0u) * cap_total] = term[b].x; t[(b * 3u + 1u) * cap_total] = term[b].y; t[(b * 3u + 2u) * cap_total] = term[b].z; }
        t[(size_t)NB * 3u * cap_total] = __uint_as_float(n_terms);
      }
    }
  }
  flush_stats<COUNT>(p, lane, R, W, 0u);
}

// PT part 3 — `indirect +=` in the order of compute.wgsl:376-404 (samples ascending, bounces ascending), then :405-409.
__global__ __launch_bounds__(TBLK) void k_pt_resolve(const RtRenderParams p) {
  const uint32_t shard = blockIdx.y, r = blockIdx.x * TBLK + threadIdx.x;
  if (r >= p.pt_count[shard]) return;
  const uint32_t GI = (uint32_t)p.gi_samples, NB = (uint32_t)p.gi_bounces;
  const size_t cap_total = (size_t)N_SHARD * p.pt_cap, g = (size_t)shard * p.pt_cap + r;
  const float4* d = p.pt_recs + g * 3;
  const float4 a = d[0], b4 = d[1], c = d[2];
  V3 indirect = v3(0.0f, 0.0f, 0.0f);
  for (uint32_t i = 0; i < GI; ++i) {
    const float* t = p.pt_terms + (size_t)i * (NB * 3u + 1u) * cap_total + g;
    const uint32_t n = __float_as_uint(t[(size_t)NB * 3u * cap_total]);
    for (uint32_t b = 0; b < n; ++b)
      indirect = indirect + v3(t[(b * 3u + 0u) * cap_total], t[(b * 3u + 1u) * cap_total], t[(b * 3u + 2u) * cap_total]);
  }
  indirect = indirect / (float)GI;
  const V3 albedo = xyz(c);
  const float d0 = a.w;
  const V3 col = v3(__fmaf_rn(albedo.x, d0, indirect.x), __fmaf_rn(albedo.y, d0, indirect.y), __fmaf_rn(albedo.z, d0, indirect.z));
  const float gm = 1.0f / 2.2f;
  if (p.rgba8)
    reinterpret_cast<uint32_t*>(p.rgba8)[__float_as_uint(b4.w)] =
        unorm8(rt_pow(col.x, gm)) | (unorm8(rt_pow(col.y, gm)) << 8) | (unorm8(rt_pow(col.z, gm)) << 16) | 0xFF000000u;
}

// surface records into BVH order (the vt / material counterpart of flattenForSSBO's triPos / triNor loop, script.js:530-545)
__global__ __launch_bounds__(TBLK) void k_gather_uv(const float* __restrict__ texcoords, const uint32_t* __restrict__ tex_idx,
                                                    const int32_t* __restrict__ material_ids, const uint32_t* __restrict__ order,
                                                    uint32_t n_tri, float4* __restrict__ tri_uv) {
  const uint32_t i = blockIdx.x * TBLK + threadIdx.x;
  if (i >= n_tri) return;
  const uint32_t t = order[i];
  const float2* tc = reinterpret_cast<const float2*>(texcoords);
  const float2 a = tc[tex_idx[3 * (size_t)t + 0]], b = tc[tex_idx[3 * (size_t)t + 1]], c = tc[tex_idx[3 * (size_t)t + 2]];
  const int32_t m = material_ids ? material_ids[t] : -1;
  tri_uv[2 * (size_t)i + 0] = make_float4(a.x, a.y, b.x, b.y);
  tri_uv[2 * (size_t)i + 1] = make_float4(c.x, c.y, __uint_as_float((uint32_t)m), 0.0f);
}

template <typename T>
__global__ __launch_bounds__(TBLK) void k_deinterleave(const T* __restrict__ packed, T* __restrict__ image, uint32_t width,
                                                       uint32_t height, int stripe_rows, int world, uint32_t rows_per_rank) {
  const size_t n = (size_t)width * height;
  for (size_t i = (size_t)blockIdx.x * TBLK + threadIdx.x; i < n; i += (size_t)gridDim.x * TBLK) {
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const uint32_t stripe = y / (uint32_t)stripe_rows, rin = y % (uint32_t)stripe_rows;
    const uint32_t rank = stripe % (uint32_t)world, sl = stripe / (uint32_t)world;
    const size_t src = ((size_t)rank * rows_per_rank + (size_t)sl * stripe_rows + rin) * width + x;
    image[i] = packed[src];
  }
}

// The frame records of a batch, 16 per launch as kernel arguments, into the batch's device array.
__global__ __launch_bounds__(64) void k_put_frames(const RtFrameChunk c, RtFrameRec* __restrict__ dst, uint32_t n) {
  constexpr uint32_t WORDS = sizeof(RtFrameRec) / 4u;
  static_assert(sizeof(RtFrameRec) % 8 == 0 && sizeof(RtFrameChunk) <= 3584, "frame records travel as kernel arguments");
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);
  for (uint32_t i = threadIdx.x; i < n * WORDS; i += 64u) d[i] = src[i];
}

// The same with WINDOWS instead of whole stripe buffers (rt_assemble_device_on): every rank sent only columns
// [col0, col0 + n_cols) of its packed rows [group0 * stripe_rows, ... + n_rows) - stripe group g = image rows
// [g * world * stripe_rows, (g + 1) * world * stripe_rows), one stripe of every rank - and everything outside is `fill`.
__global__ __launch_bounds__(TBLK) void k_assemble(const uint32_t* __restrict__ windows, uint32_t* __restrict__ image, uint32_t width,
                                                   uint32_t height, uint32_t stripe_rows, uint32_t world, uint32_t col0, uint32_t n_cols,
                                                   uint32_t group0, uint32_t n_rows, size_t rank_stride, uint32_t fill) {
  const size_t n = (size_t)width * height;
  for (size_t i = (size_t)blockIdx.x * TBLK + threadIdx.x; i < n; i += (size_t)gridDim.x * TBLK) {
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const uint32_t stripe = y / stripe_rows, rin = y % stripe_rows;
    const uint32_t rank = stripe % world, grp = stripe / world;
    const uint32_t wr = (grp - group0) * stripe_rows + rin;              // (wraps to a huge value when grp < group0)
    uint32_t v = fill;
    if (grp >= group0 && wr < n_rows && x >= col0 && x - col0 < n_cols) v = windows[(size_t)rank * rank_stride + (size_t)wr * n_cols + (x - col0)];
    image[i] = v;
  }
}

}  // namespace

// Persistent grid of phase B: what is actually resident (occupancy query), so that no block starts
// late behind another one (measured: with a 4-blocks/CU grid and 3 resident, a quarter of the
// waves started 400 us into a 650 us launch).
uint32_t rt_render_grid(int n_cu) { return (uint32_t)n_cu * 16u; }   // upper bound (sizes the per-block stats slots: two launches with the PT split)
// The k_render instantiation a frame runs.  Every launch and every occupancy query goes through this one table, so the
// persistent grid is sized from the kernel that is actually launched (COUNT = false variants hold fewer registers).
template <int L, bool C, bool X, int K, bool BT>
static const void* render_kernel_ptr() {
  if constexpr (BT) return reinterpret_cast<const void*>(&k_render<L, C, X, K, true, RtBatch>);
  else return reinterpret_cast<const void*>(&k_render<L, C, X, K, false, RtNoBatch>);
}
template <int K, bool BT>
static const void* render_kernel_lcx(int layout, bool count, bool tex) {
  // (counting launches are not timed: they keep the 64-bit form, one instantiation less each)
  if (layout == RT_LAYOUT_WIDE64_NEAR && !count) return tex ? render_kernel_ptr<RT_LAYOUT_WIDE64_NEAR, false, true, K, BT>() : render_kernel_ptr<RT_LAYOUT_WIDE64_NEAR, false, false, K, BT>();
  if (layout != RT_LAYOUT_AOS32) {
    if (count) return tex ? render_kernel_ptr<RT_LAYOUT_WIDE64, true, true, K, BT>() : render_kernel_ptr<RT_LAYOUT_WIDE64, true, false, K, BT>();
    return tex ? render_kernel_ptr<RT_LAYOUT_WIDE64, false, true, K, BT>() : render_kernel_ptr<RT_LAYOUT_WIDE64, false, false, K, BT>();
  }
  if (count) return tex ? render_kernel_ptr<RT_LAYOUT_AOS32, true, true, K, BT>() : render_kernel_ptr<RT_LAYOUT_AOS32, true, false, K, BT>();
  return tex ? render_kernel_ptr<RT_LAYOUT_AOS32, false, true, K, BT>() : render_kernel_ptr<RT_LAYOUT_AOS32, false, false, K, BT>();
}
// frag kinds: WIDE64 records, constant albedo (rt_render_device rejects anything else); a batch: the pixel kind only
static const void* render_kernel(int layout, bool count, bool tex, int kind, bool batch) {
  if (kind == KIND_FRAG_WEB && batch) return count ? render_kernel_ptr<RT_LAYOUT_WIDE64, true, false, KIND_FRAG_WEB, true>() : render_kernel_ptr<RT_LAYOUT_WIDE64, false, false, KIND_FRAG_WEB, true>();
  if (kind == KIND_FRAG_ROOT && batch) return count ? render_kernel_ptr<RT_LAYOUT_WIDE64, true, false, KIND_FRAG_ROOT, true>() : render_kernel_ptr<RT_LAYOUT_WIDE64, false, false, KIND_FRAG_ROOT, true>();
  if (kind == KIND_FRAG_WEB) return count ? render_kernel_ptr<RT_LAYOUT_WIDE64, true, false, KIND_FRAG_WEB, false>() : render_kernel_ptr<RT_LAYOUT_WIDE64, false, false, KIND_FRAG_WEB, false>();
  if (kind == KIND_FRAG_ROOT) return count ? render_kernel_ptr<RT_LAYOUT_WIDE64, true, false, KIND_FRAG_ROOT, false>() : render_kernel_ptr<RT_LAYOUT_WIDE64, false, false, KIND_FRAG_ROOT, false>();
  if (kind == KIND_PTSPLIT) return render_kernel_lcx<KIND_PTSPLIT, false>(layout, count, tex);
  if (kind == KIND_TRAVERSE) return batch ? render_kernel_lcx<KIND_TRAVERSE, true>(layout, count, false) : render_kernel_lcx<KIND_TRAVERSE, false>(layout, count, false);
  if (kind == KIND_SHADE) return batch ? render_kernel_lcx<KIND_SHADE, true>(layout, count, tex) : render_kernel_lcx<KIND_SHADE, false>(layout, count, tex);
  if (kind == KIND_PIXEL_NOGI && batch) return render_kernel_lcx<KIND_PIXEL_NOGI, true>(layout, count, tex);
  return batch ? render_kernel_lcx<KIND_PIXEL, true>(layout, count, tex) : render_kernel_lcx<KIND_PIXEL, false>(layout, count, tex);
}
static uint32_t resident_grid(int n_cu, int max_stack, const void* kernel, int cap = 0, const char* env = "RTCORE_BLOCKS_PER_CU") {
  struct Slot { const void* k; int stack; int per_cu; };
  static Slot slots[64];
  static int n_slots = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int per_cu = 0;
  for (int i = 0; i < n_slots; ++i) if (slots[i].k == kernel && slots[i].stack == max_stack) per_cu = slots[i].per_cu;
  if (per_cu == 0) {
    int nb = 0;
    const size_t lds = (size_t)max_stack * TBLK * 8;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, TBLK, lds);
    if (e != hipSuccess || nb < 1) nb = 2;
    per_cu = nb > 6 ? 6 : nb;
    if (n_slots < 64) slots[n_slots++] = Slot{kernel, max_stack, per_cu};
  }
  int want = (cap > 0 && cap < per_cu) ? cap : per_cu;      // per_cu = what is resident: never more than that
  if (const char* e = getenv(env)) { const int v = atoi(e); if (v >= 1) want = v < per_cu ? v : per_cu; }   // (experiments: read on every launch; may lift a cap)
  return (uint32_t)n_cu * (uint32_t)want;
}
uint32_t rt_cull_blocks(uint32_t n_tiles) { return (n_tiles + CULL_TPB - 1) / CULL_TPB; }
uint32_t rt_n_class() { return (uint32_t)N_CLASS; }
uint32_t rt_shard_words() { return N_SHARD * (uint32_t)shard_stride_words(); }
uint32_t rt_shard_stride() { return (uint32_t)shard_stride_words(); }
// a shard owns at most ceil(n_chunks / (N_SHARD * group)) groups of `group` chunks of 64 pixels
uint32_t rt_pt_capacity(uint32_t n_pixels, uint32_t group) {
  const uint64_t chunks = ((uint64_t)n_pixels + 63u) / 64u, per = (uint64_t)N_SHARD * group;
  return (uint32_t)(((chunks + per - 1u) / per) * group * 64u);
}

// k_cull zeroes shard_heads; masks / block_total / list / list_count / stats are fully rewritten.
// Screen-space rectangle of the root box.  A primary ray through pixel-plane point s = (sx, sy, 1) has
// object-space direction A s, A = M3 [camRight camUp camForward] (primary_ray above), origin o = M (cam, 1).
// A box point c reached at t >= 0 satisfies A^-1 (c - o) = lambda (sx, sy, 1) with lambda > 0, so when all 8
// corners have lambda > 0 every entering ray's (sx, sy) lies in the convex hull of the corner projections,
// hence in their bounding rectangle.  Two pixels of padding absorb f32 rounding of the device's ray set-up
// (< 1e-3 pixel).  Anything else (camera inside / beside the box, singular A, NaN) keeps the full image.
void rt_cull_rect(RtRenderParams& p, const float rb[8]) {
  p.cull_x0 = 0; p.cull_y0 = 0; p.cull_x1 = (int)p.width - 1; p.cull_y1 = (int)p.height - 1;
  const float* U = p.U;
  const float* m = U + 20;
  double A[3][3], o[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      const float* v = U + 8 + 4 * c;   // camRight @8, camUp @12, camForward @16
      A[r][c] = (double)m[r] * v[0] + (double)m[4 + r] * v[1] + (double)m[8 + r] * v[2];
    }
    o[r] = (double)m[r] * U[0] + (double)m[4 + r] * U[1] + (double)m[8 + r] * U[2] + (double)m[12 + r];
  }
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  double amax = 0.0;
  for (auto& row : A) for (double v : row) amax = std::fmax(amax, std::fabs(v));
  if (!(std::fabs(det) > 1e-12 * amax * amax * amax) || !(amax > 0.0)) return;
  double inv[3][3];
  inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
  inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det; inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
  inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
  inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
  inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
  double sx0 = 1e300, sx1 = -1e300, sy0 = 1e300, sy1 = -1e300;
  for (int k = 0; k < 8; ++k) {
    const double c[3] = {(double)rb[(k & 1) ? 4 : 0] - o[0], (double)rb[(k & 2) ? 5 : 1] - o[1], (double)rb[(k & 4) ? 6 : 2] - o[2]};
    double q[3];
    for (int r = 0; r < 3; ++r) q[r] = inv[r][0] * c[0] + inv[r][1] * c[1] + inv[r][2] * c[2];
    const double qm = std::fmax(std::fabs(q[0]), std::fmax(std::fabs(q[1]), std::fabs(q[2])));
    if (!(q[2] > 1e-6 * qm) || !(qm < 1e300)) return;                     // corner beside / behind the eye, or NaN
    sx0 = std::fmin(sx0, q[0] / q[2]); sx1 = std::fmax(sx1, q[0] / q[2]);
    sy0 = std::fmin(sy0, q[1] / q[2]); sy1 = std::fmax(sy1, q[1] / q[2]);
  }
  // sx = 2 (x + 0.5) / w - 1  =>  x = (sx + 1) w / 2 - 0.5
  const double W = (double)p.width, H = (double)p.height, pad = 2.0;
  const double x0 = std::floor((sx0 + 1.0) * 0.5 * W - 0.5 - pad), x1 = std::ceil((sx1 + 1.0) * 0.5 * W - 0.5 + pad);
  const double y0 = std::floor((sy0 + 1.0) * 0.5 * H - 0.5 - pad), y1 = std::ceil((sy1 + 1.0) * 0.5 * H - 0.5 + pad);
  if (!(x0 == x0) || !(x1 == x1) || !(y0 == y0) || !(y1 == y1)) return;
  p.cull_x0 = (int)std::fmax(0.0, std::fmin(x0, W));  p.cull_x1 = (int)std::fmin(W - 1.0, std::fmax(x1, -1.0));
  p.cull_y0 = (int)std::fmax(0.0, std::fmin(y0, H));  p.cull_y1 = (int)std::fmin(H - 1.0, std::fmax(y1, -1.0));
}

void rt_launch_render(const RtRenderParams& p_in, int layout, bool count_work, bool want_stats, int n_cu, hipStream_t s, hipEvent_t ev0,
                      hipEvent_t ev1, hipEvent_t ev_mid) {
  RtRenderParams p = p_in;
  const dim3 b(TBLK), gc(rt_cull_blocks(p.n_tiles));
  const bool tex = p.tri_uv != nullptr, split = p.pt_cap != 0;
  const int kind = p.frag == 1 ? KIND_FRAG_WEB : p.frag == 2 ? KIND_FRAG_ROOT : split ? KIND_PTSPLIT : KIND_PIXEL;
  if (p.frag) layout = RT_LAYOUT_WIDE64;
  // RT_MODE_PRIMARY / SHADOW with a hit buffer: two launches, the primary rays' walk and the colour of the hits
  const bool two_phase = p.hits != nullptr && kind == KIND_PIXEL && p.mode != RT_MODE_PT;
  const int klayout = (layout == RT_LAYOUT_WIDE64 && p.near32 && !p.frag) ? RT_LAYOUT_WIDE64_NEAR : layout;   // k_render's instantiation; k_pt_samples keeps `layout`
  const void* kernel = two_phase ? render_kernel(klayout, count_work, false, KIND_TRAVERSE, false) : render_kernel(klayout, count_work, tex && !p.frag, kind, false);
  // one frame per launch: more resident waves stretch the tail (every chain of dependent fetches gets slower) and lose what
  // they hide; 3 blocks per CU stay the default there (profiles/r3/ab_two_phase.txt), a batch takes all that fit.  (Built without
  // the SLP vectoriser the untextured whole-pixel kernel fits 4: cfg 3 untextured, one frame per pipeline, 2 / 3 / 4 blocks per CU
  // = 0.506 / 0.485 / 0.510 ms, primary only 0.482 / 0.466 / 0.504; the PT split's launches do best with all that fit.)
  const int one_frame_cap = (two_phase || kind == KIND_PIXEL) ? 3 : 0;
  p.render_blocks = resident_grid(n_cu, p.max_stack, kernel, one_frame_cap, two_phase ? "RTCORE_TRAVERSE_BPC" : "RTCORE_BLOCKS_PER_CU");
  // A persistent grid that fills the chip leaves a frame queued behind it nothing but its tail to overlap with.  Frames
  // in flight therefore take 2 of the (usually 3) resident blocks per CU each, so that two of them really run side by
  // side (cfg 3, two in flight: 0.440 -> 0.400 ms per frame; a 1/8 stripe share, three in flight: 0.146 -> 0.125 ms).
  if (p.blocks_per_cu_cap && !getenv("RTCORE_BLOCKS_PER_CU")) p.render_blocks = std::min(p.render_blocks, (uint32_t)n_cu * p.blocks_per_cu_cap);
  RtNoBatch nb;
  hipLaunchKernelGGL((k_cull<false, RtNoBatch>), gc, b, 0, s, p, nb);
  hipLaunchKernelGGL((k_scan<false, RtNoBatch>), dim3(1), b, 0, s, p, nb);
  hipLaunchKernelGGL((k_compact<false, RtNoBatch>), gc, b, 0, s, p, nb);
  const dim3 g(p.render_blocks);
  const size_t lds = (size_t)p.max_stack * TBLK * (sizeof(uint32_t) + sizeof(float));
  if (ev0) (void)hipEventRecord(ev0, s);
  void* args[2] = {&p, &nb};
  (void)hipLaunchKernel(kernel, g, b, args, lds, s);
  if (ev_mid) (void)hipEventRecord(ev_mid, s);          // behind the first traversal launch (the only one unless the frame is split)
  uint32_t stat_blocks = split ? 2u * g.x : g.x;
  if (two_phase) {
    const void* shade = render_kernel(klayout, count_work, tex, KIND_SHADE, false);
    RtRenderParams p2 = p;
    p2.render_blocks = resident_grid(n_cu, 0, shade, 4, "RTCORE_SHADE_BPC");
    p2.stats_partial = p.stats_partial + (size_t)g.x * 8;
    void* args2[2] = {&p2, &nb};
    (void)hipLaunchKernel(shade, dim3(p2.render_blocks), b, args2, 0, s);
    stat_blocks = g.x + p2.render_blocks;
  }
  if (split) {   // PT split: part 1 (KIND_PTSPLIT) then part 2 (k_pt_samples) with its own rows of the per-block stats partials
    RtRenderParams p2 = p;
    p2.stats_partial = p.stats_partial + (size_t)g.x * 8;
#define RT_LAUNCH_PT2(L, C, X) hipLaunchKernelGGL((k_pt_samples<L, C, X>), g, b, lds, s, p2)
    if (layout == RT_LAYOUT_WIDE64) {
      if (count_work) { if (tex) RT_LAUNCH_PT2(RT_LAYOUT_WIDE64, true, true); else RT_LAUNCH_PT2(RT_LAYOUT_WIDE64, true, false); }
      else { if (tex) RT_LAUNCH_PT2(RT_LAYOUT_WIDE64, false, true); else RT_LAUNCH_PT2(RT_LAYOUT_WIDE64, false, false); }
    } else {
      if (count_work) { if (tex) RT_LAUNCH_PT2(RT_LAYOUT_AOS32, true, true); else RT_LAUNCH_PT2(RT_LAYOUT_AOS32, true, false); }
      else { if (tex) RT_LAUNCH_PT2(RT_LAYOUT_AOS32, false, true); else RT_LAUNCH_PT2(RT_LAYOUT_AOS32, false, false); }
    }
#undef RT_LAUNCH_PT2
    hipLaunchKernelGGL(k_pt_resolve, dim3((p.pt_cap + TBLK - 1) / TBLK, N_SHARD), b, 0, s, p);
  }
  if (ev1) (void)hipEventRecord(ev1, s);
  if (want_stats) hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, s, p, stat_blocks, p.local_rows * p.width, count_work ? 1 : 0);
}

// A batch: the frames' cull / scan / compact passes as one grid dimension of three launches, then ONE persistent k_render over all
// the frames' lists (rt_render_batch_device).  p carries what the frames share (tree, geometry of the image, scratch
// bases); bt what differs.  No stats, no PT split.
void rt_launch_render_batch(const RtRenderParams& p_in, const RtBatch& bt, int layout, bool count_work, int n_cu, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                            hipEvent_t ev_mid) {
  RtRenderParams p = p_in;
  RtBatch batch = bt;
  const dim3 b(TBLK), gc(rt_cull_blocks(p.n_tiles), bt.n);
  const bool tex = p.tri_uv != nullptr;
  const bool two_phase = p.hits != nullptr && p.mode != RT_MODE_PT && !p.frag;      // as rt_launch_render
  const int kind = p.frag == 1 ? KIND_FRAG_WEB : p.frag == 2 ? KIND_FRAG_ROOT : p.mode == RT_MODE_PT ? KIND_PIXEL : KIND_PIXEL_NOGI;
  if (p.frag) layout = RT_LAYOUT_WIDE64;
  const int klayout = (layout == RT_LAYOUT_WIDE64 && p.near32 && !p.frag) ? RT_LAYOUT_WIDE64_NEAR : layout;
  const void* kernel = two_phase ? render_kernel(klayout, count_work, false, KIND_TRAVERSE, true) : render_kernel(klayout, count_work, tex && !p.frag, kind, true);
  // the traversal launch: 5 blocks per CU (of the 6 that fit) - 4 / 5 / 6: 0.254 / 0.234 / 0.239 ms per frame with the shade launch at 4
  p.render_blocks = resident_grid(n_cu, p.max_stack, kernel, two_phase ? 5 : 0, two_phase ? "RTCORE_TRAVERSE_BPC" : "RTCORE_BLOCKS_PER_CU");
  if (p.blocks_per_cu_cap && !getenv("RTCORE_BLOCKS_PER_CU")) p.render_blocks = std::min(p.render_blocks, (uint32_t)n_cu * p.blocks_per_cu_cap);   // pipelines in flight share the CUs
  hipLaunchKernelGGL((k_cull<true, RtBatch>), gc, b, 0, s, p, batch);
  hipLaunchKernelGGL((k_scan<true, RtBatch>), dim3(bt.n), b, 0, s, p, batch);
  hipLaunchKernelGGL((k_compact<true, RtBatch>), gc, b, 0, s, p, batch);
  const size_t lds = (size_t)p.max_stack * TBLK * (sizeof(uint32_t) + sizeof(float));
  const uint32_t slots_a = p.render_blocks * (TBLK / 64);
  uint32_t slots_b = 0;
  RtRenderParams p2 = p;
  const void* shade = nullptr;
  if (two_phase) {
    shade = render_kernel(klayout, count_work, tex, KIND_SHADE, true);
    p2.render_blocks = resident_grid(n_cu, 0, shade, 5, "RTCORE_SHADE_BPC");
    if (p.blocks_per_cu_cap && !getenv("RTCORE_SHADE_BPC")) p2.render_blocks = std::min(p2.render_blocks, (uint32_t)n_cu * p.blocks_per_cu_cap);
    slots_b = p2.render_blocks * (TBLK / 64);
    p2.stats_partial = p.stats_partial ? p.stats_partial + (size_t)8 * bt.n * slots_a : nullptr;   // the shade phase's wave slots lie behind the traversal's
  }
  if (count_work) (void)hipMemsetAsync(p.stats_partial, 0, sizeof(unsigned long long) * 8 * (size_t)bt.n * (slots_a + slots_b), s);
  if (ev0) (void)hipEventRecord(ev0, s);
  void* args[2] = {&p, &batch};
  (void)hipLaunchKernel(kernel, dim3(p.render_blocks), b, args, lds, s);
  if (ev_mid) (void)hipEventRecord(ev_mid, s);
  if (two_phase) {
    void* args2[2] = {&p2, &batch};
    (void)hipLaunchKernel(shade, dim3(p2.render_blocks), b, args2, 0, s);
  }
  if (ev1) (void)hipEventRecord(ev1, s);
  if (count_work) hipLaunchKernelGGL(k_stats_final_batch, dim3(bt.n), dim3(256), 0, s, p, batch, slots_a, slots_b, p.local_rows * p.width, 1);
}

void rt_launch_gather_uv(const float* texcoords, const uint32_t* tex_idx, const int32_t* material_ids, const uint32_t* order,
                         uint32_t n_tri, float* tri_uv, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_uv, dim3((n_tri + TBLK - 1) / TBLK), dim3(TBLK), 0, s, texcoords, tex_idx, material_ids, order, n_tri,
                     reinterpret_cast<float4*>(tri_uv));
}

void rt_launch_put_frames(const RtFrameRec* host_recs, uint32_t n, RtFrameRec* d_frames, hipStream_t s) {
  for (uint32_t f0 = 0; f0 < n; f0 += RT_FRAMES_PER_PUT) {
    RtFrameChunk c;
    const uint32_t k = std::min<uint32_t>(RT_FRAMES_PER_PUT, n - f0);
    std::memset(&c, 0, sizeof(c));
    std::memcpy(c.r, host_recs + f0, sizeof(RtFrameRec) * k);
    hipLaunchKernelGGL(k_put_frames, dim3(1), dim3(64), 0, s, c, d_frames + f0, k);
  }
}

void rt_launch_assemble(const void* windows, void* image, uint32_t width, uint32_t height, int stripe_rows, int world, uint32_t col0,
                        uint32_t n_cols, uint32_t group0, uint32_t n_rows, size_t rank_stride, uint32_t fill, hipStream_t s) {
  const size_t n = (size_t)width * height;
  int blocks = (int)((n + TBLK - 1) / TBLK);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(TBLK), 0, s, (const uint32_t*)windows, (uint32_t*)image, width, height,
                     (uint32_t)stripe_rows, (uint32_t)world, col0, n_cols, group0, n_rows, rank_stride, fill);
}

void rt_launch_deinterleave(const void* packed, void* image, uint32_t width, uint32_t height, uint32_t elem_bytes,
                            int stripe_rows, int world, uint32_t rows_per_rank, hipStream_t s) {
  const size_t n = (size_t)width * height;
  int blocks = (int)((n + TBLK - 1) / TBLK);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (elem_bytes == 4)
    hipLaunchKernelGGL((k_deinterleave<uint32_t>), dim3(blocks), dim3(TBLK), 0, s, (const uint32_t*)packed, (uint32_t*)image,
                       width, height, stripe_rows, world, rows_per_rank);
  else
    hipLaunchKernelGGL((k_deinterleave<uint8_t>), dim3(blocks), dim3(TBLK), 0, s, (const uint8_t*)packed, (uint8_t*)image,
                       width * elem_bytes, height, stripe_rows, world, rows_per_rank);
}
