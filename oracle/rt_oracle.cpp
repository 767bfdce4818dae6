// rt_oracle.cpp — scalar CPU ORACLE (test infrastructure; see rt_oracle.h header comment).
// PARITY UNPINNED BY THE REFERENCE (no reference tests/fixtures exist):
// pinned by hand-derived known answers + property tests under tests/.
//
// Every function cites the reference lines it restates (paths under /root/reference).
// Float rules (SURVEY.md §7 "hard parts"): strict IEEE f32, no contraction
// (-ffp-contract=off), fma only where the WGSL writes fma, left-to-right sums,
// correctly rounded / and sqrt, fmin/fmax NaN semantics, saturating float->int.
#include "rt_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- vec helpers
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// WGSL dot / cross, unfused, left to right
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline V3 fma3(V3 a, float s, V3 c) {  // WGSL fma(a, vec3(s), c)
  return v3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z));
}
inline V3 fma3v(V3 a, V3 b, V3 c) { return v3(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)); }
inline V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

// WGSL f32 -> u32 / i32 conversions saturate; NaN -> 0.
inline uint32_t sat_u32(float f) {
  if (!(f > 0.0f)) return 0u;              // negatives, -0, NaN
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
inline int32_t sat_i32(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)f;
}

// total-order min/max used for the centroid bounds (−0 < +0) so that the GPU's
// ordered-integer atomics give the identical bit pattern.
inline float tmin(float a, float b) { return (a < b || (a == b && std::signbit(a))) ? a : b; }
inline float tmax(float a, float b) { return (a > b || (a == b && !std::signbit(a))) ? a : b; }

}  // namespace

// ============================================================ Morton (hot path 1)
// web/morton_compute.wgsl:18-25
extern "C" uint32_t orc_expand_bits(uint32_t v) {
  uint32_t x = v & 0x3FFu;
  x = (x * 0x00010001u) & 0xFF0000FFu;
  x = (x * 0x00000101u) & 0x0F00F00Fu;
  x = (x * 0x00000011u) & 0xC30C30C3u;
  x = (x * 0x00000005u) & 0x49249249u;
  return x;
}
// web/morton_compute.wgsl:28-31 ; NaN pinned to 0 (fmaxf returns the non-NaN operand)
extern "C" uint32_t orc_quantize(float x) {
  float f = floorf(x * 1024.0f);
  f = fminf(fmaxf(f, 0.0f), 1023.0f);
  return sat_u32(f);
}

namespace {
inline V3 load3(const float* p, uint32_t i) { return v3(p[3 * i + 0], p[3 * i + 1], p[3 * i + 2]); }
// web/morton_compute.wgsl:52 : (p0 + p1 + p2) / 3.0
inline V3 centroid(const float* pos, const uint32_t* vidx, uint32_t t) {
  V3 p0 = load3(pos, vidx[3 * t + 0]), p1 = load3(pos, vidx[3 * t + 1]), p2 = load3(pos, vidx[3 * t + 2]);
  return ((p0 + p1) + p2) / 3.0f;
}
}  // namespace

// the reference never fills `Bounds`; defined as centroid min / reciprocal extent.
extern "C" void orc_centroid_bounds(const float* pos, const uint32_t* vidx, uint32_t n_tri,
                                    float bmin[3], float inv_extent[3]) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t t = 0; t < n_tri; ++t) {
    V3 c = centroid(pos, vidx, t);
    float cc[3] = {c.x, c.y, c.z};
    for (int k = 0; k < 3; ++k) { mn[k] = tmin(mn[k], cc[k]); mx[k] = tmax(mx[k], cc[k]); }
  }
  for (int k = 0; k < 3; ++k) {
    float ext = mx[k] - mn[k];
    bmin[k] = mn[k];
    inv_extent[k] = (ext > 0.0f) ? (1.0f / ext) : 0.0f;
  }
}

// web/morton_compute.wgsl:33-69
extern "C" void orc_morton(const float* pos, const uint32_t* vidx, uint32_t n_tri,
                           const float bmin[3], const float inv_extent[3], uint32_t* codes) {
  V3 bm = v3(bmin[0], bmin[1], bmin[2]), ie = v3(inv_extent[0], inv_extent[1], inv_extent[2]);
  for (uint32_t t = 0; t < n_tri; ++t) {
    V3 ctr = centroid(pos, vidx, t);
    V3 nrm = (ctr - bm) * ie;                                   // :55 separate sub and mul
    uint32_t xx = orc_expand_bits(orc_quantize(nrm.x));
    uint32_t yy = orc_expand_bits(orc_quantize(nrm.y));
    uint32_t zz = orc_expand_bits(orc_quantize(nrm.z));
    codes[t] = (xx << 2) | (yy << 1) | zz;                     // :68
  }
}

// stable sort => equal codes keep ascending original triangle index.
extern "C" void orc_sort(const uint32_t* codes, uint32_t n_tri, uint32_t* sorted_codes, uint32_t* order) {
  std::vector<uint32_t> idx(n_tri);
  for (uint32_t i = 0; i < n_tri; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return codes[a] < codes[b]; });
  for (uint32_t i = 0; i < n_tri; ++i) { order[i] = idx[i]; sorted_codes[i] = codes[idx[i]]; }
}

namespace {
// Karras 2012 delta: common-prefix length of the augmented keys (code, index); -1 out of range.
inline int delta(const uint32_t* k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  uint32_t a = k[i], b = k[j];
  if (a != b) return __builtin_clz(a ^ b);
  return 32 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}
}  // namespace

extern "C" void orc_karras(const uint32_t* k, uint32_t n_tri, int32_t* child0, int32_t* child1, int32_t* parent) {
  const int n = (int)n_tri;
  if (n == 0) return;
  parent[0] = -1;
  for (int i = 0; i < n - 1; ++i) {
    int d = (delta(k, n, i, i + 1) - delta(k, n, i, i - 1)) > 0 ? 1 : -1;
    int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (delta(k, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(k, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
      if (delta(k, n, i, i + (s + t) * d) > dnode) s += t;
      if (t == 1) break;
    }
    int gamma = i + s * d + std::min(d, 0);
    int lo = std::min(i, j), hi = std::max(i, j);
    int c0 = (lo == gamma) ? (n - 1) + gamma : gamma;
    int c1 = (hi == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    child0[i] = c0; child1[i] = c1;
    parent[c0] = i; parent[c1] = i;
  }
}

// bounds -> morton -> sort -> tree -> refit -> flat buffers
// Node slots A (float 3) and B (float 7) of the 32-byte record:
//   v1 (web/script.js:546-559): indices as f32 VALUES, leaf: A = tri_start, B = -count (< 0)  — exact below 2^24 (F11)
//   v2:           u32 bit-casts in the same slots, leaf flag = bit 31 of B, low 31 bits = count
static inline float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float enc_index(int enc, uint32_t i) { return enc == 2 ? bits_f(i) : (float)i; }
static inline float enc_leaf_b(int enc, uint32_t count) { return enc == 2 ? bits_f(0x80000000u | count) : -(float)count; }

extern "C" void orc_build_lbvh(const float* pos, uint32_t n_vert, const float* nor, uint32_t n_nor,
                               const uint32_t* vidx, const uint32_t* nidx, uint32_t n_tri,
                               float* nodes, float* tri_pos, float* tri_nor,
                               uint32_t* codes_out, uint32_t* order_out, double* seconds) {
  orc_build_lbvh_enc(pos, n_vert, nor, n_nor, vidx, nidx, n_tri, nodes, tri_pos, tri_nor, codes_out, order_out, seconds, 1);
}

extern "C" void orc_build_lbvh_enc(const float* pos, uint32_t n_vert, const float* nor, uint32_t n_nor,
                                   const uint32_t* vidx, const uint32_t* nidx, uint32_t n_tri,
                                   float* nodes, float* tri_pos, float* tri_nor,
                                   uint32_t* codes_out, uint32_t* order_out, double* seconds, int enc) {
  (void)n_vert; (void)n_nor;
  auto t0 = std::chrono::steady_clock::now();
  const uint32_t T = n_tri;
  if (T == 0) { if (seconds) *seconds = 0; return; }
  float bmin[3], inv[3];
  orc_centroid_bounds(pos, vidx, T, bmin, inv);
  std::vector<uint32_t> codes(T), sorted(T), order(T);
  orc_morton(pos, vidx, T, bmin, inv, codes.data());
  orc_sort(codes.data(), T, sorted.data(), order.data());
  const uint32_t N = 2 * T - 1;
  std::vector<int32_t> c0(T > 1 ? T - 1 : 1), c1(T > 1 ? T - 1 : 1), par(N);
  orc_karras(sorted.data(), T, c0.data(), c1.data(), par.data());

  // triangle buffers in BVH order — web/script.js:527-545
  for (uint32_t i = 0; i < T; ++i) {
    uint32_t ti = order[i];
    for (int v = 0; v < 3; ++v) {
      const size_t vi = vidx[3 * (size_t)ti + v], ni = nidx[3 * (size_t)ti + v];
      float* dp = tri_pos + ((size_t)i * 3 + v) * 4;
      float* dn = tri_nor + ((size_t)i * 3 + v) * 4;
      dp[0] = pos[3 * vi + 0]; dp[1] = pos[3 * vi + 1]; dp[2] = pos[3 * vi + 2]; dp[3] = 0.0f;
      dn[0] = nor[3 * ni + 0]; dn[1] = nor[3 * ni + 1]; dn[2] = nor[3 * ni + 2]; dn[3] = 0.0f;
    }
  }
  // leaves: true triangle bounds (web/script.js:309-316); leaf j -> node (T-1)+j,
  // A = tri_start (BVH-order index), B = -count  (web/script.js:348,554,558)
  for (uint32_t j = 0; j < T; ++j) {
    const float* p = tri_pos + (size_t)j * 12;
    float* nd = nodes + (size_t)((T - 1) + j) * 8;
    for (int k = 0; k < 3; ++k) {
      nd[k] = fminf(fminf(p[k], p[4 + k]), p[8 + k]);
      nd[4 + k] = fmaxf(fmaxf(p[k], p[4 + k]), p[8 + k]);
    }
    nd[3] = enc_index(enc, j);
    nd[7] = enc_leaf_b(enc, 1u);
  }
  // internal boxes bottom-up: process nodes in decreasing depth via an explicit post-order.
  if (T > 1) {
    std::vector<int32_t> stack; stack.reserve(128);
    std::vector<uint8_t> state(T - 1, 0);
    stack.push_back(0);
    while (!stack.empty()) {
      int32_t n = stack.back();
      if (state[n] == 0) {
        state[n] = 1;
        if (c1[n] < (int32_t)(T - 1)) stack.push_back(c1[n]);
        if (c0[n] < (int32_t)(T - 1)) stack.push_back(c0[n]);
      } else {
        stack.pop_back();
        const float* a = nodes + (size_t)c0[n] * 8;
        const float* b = nodes + (size_t)c1[n] * 8;
        float* nd = nodes + (size_t)n * 8;
        for (int k = 0; k < 3; ++k) { nd[k] = fminf(a[k], b[k]); nd[4 + k] = fmaxf(a[4 + k], b[4 + k]); }
        nd[3] = enc_index(enc, (uint32_t)c0[n]);   // v1: indices stored as f32 VALUES (F11; web/script.js:554,558)
        nd[7] = enc_index(enc, (uint32_t)c1[n]);
      }
    }
  }
  if (codes_out) std::memcpy(codes_out, codes.data(), sizeof(uint32_t) * T);
  if (order_out) std::memcpy(order_out, order.data(), sizeof(uint32_t) * T);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ============================================================ host-side numbers
// web/script.js:207-227 — double math, stored back to f32
extern "C" void orc_normalize_mesh(float* p, uint32_t n) {
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      double v = p[3 * i + k];
      mn[k] = std::fmin(mn[k], v); mx[k] = std::fmax(mx[k], v);
    }
  double ctr[3];
  for (int k = 0; k < 3; ++k) ctr[k] = (mn[k] + mx[k]) * 0.5;
  double sc = 1.0 / std::fmax(std::fmax(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
  for (uint32_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) p[3 * i + k] = (float)(((double)p[3 * i + k] - ctr[k]) * sc);
}

namespace {
// web/script.js:30-75 — returns Float32Array
void make_trs(const double rot[3], const double scale[3], const double pos[3], float out[16]) {
  double cx = std::cos(rot[0]), sx_ = std::sin(rot[0]);
  double cy = std::cos(rot[1]), sy_ = std::sin(rot[1]);
  double cz = std::cos(rot[2]), sz_ = std::sin(rot[2]);
  double m00 = cz * cy, m01 = cz * sy_ * sx_ - sz_ * cx, m02 = cz * sy_ * cx + sz_ * sx_;
  double m10 = sz_ * cy, m11 = sz_ * sy_ * sx_ + cz * cx, m12 = sz_ * sy_ * cx - cz * sx_;
  double m20 = -sy_, m21 = cy * sx_, m22 = cy * cx;
  double sx = scale[0], sy = scale[1], sz = scale[2];
  double m[16] = {m00 * sx, m01 * sy, m02 * sz, 0, m10 * sx, m11 * sy, m12 * sz, 0,
                  m20 * sx, m21 * sy, m22 * sz, 0, pos[0], pos[1], pos[2], 1};
  for (int i = 0; i < 16; ++i) out[i] = (float)m[i];
}
// web/script.js:77-133 — reads f32, computes in double, stores f32; det==0 leaves zeros
void invert4(float out[16], const float m[16]) {
  double a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3], a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
  double a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11], a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
  double b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
  double b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
  double b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
  double b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
  double det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
  for (int i = 0; i < 16; ++i) out[i] = 0.0f;
  if (det == 0.0 || det != det) return;
  det = 1.0 / det;
  double o[16];
  o[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det;   o[1] = (-a01 * b11 + a02 * b10 - a03 * b09) * det;
  o[2] = (a31 * b05 - a32 * b04 + a33 * b03) * det;   o[3] = (-a21 * b05 + a22 * b04 - a23 * b03) * det;
  o[4] = (-a10 * b11 + a12 * b08 - a13 * b07) * det;  o[5] = (a00 * b11 - a02 * b08 + a03 * b07) * det;
  o[6] = (-a30 * b05 + a32 * b02 - a33 * b01) * det;  o[7] = (a20 * b05 - a22 * b02 + a23 * b01) * det;
  o[8] = (a10 * b10 - a11 * b08 + a13 * b06) * det;   o[9] = (-a00 * b10 + a01 * b08 - a03 * b06) * det;
  o[10] = (a30 * b04 - a31 * b02 + a33 * b00) * det;  o[11] = (-a20 * b04 + a21 * b02 - a23 * b00) * det;
  o[12] = (-a10 * b09 + a11 * b07 - a12 * b06) * det; o[13] = (a00 * b09 - a01 * b07 + a02 * b06) * det;
  o[14] = (-a30 * b03 + a31 * b01 - a32 * b00) * det; o[15] = (a20 * b03 - a21 * b01 + a22 * b00) * det;
  for (int i = 0; i < 16; ++i) out[i] = (float)o[i];
}
void normalize_js(double v[3]) {  // web/script.js:678-681 : Math.hypot(...v) || 1
  double l = std::hypot(v[0], v[1], v[2]);
  if (l == 0.0 || l != l) l = 1.0;
  for (int i = 0; i < 3; ++i) v[i] /= l;
}
void cross_js(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
}  // namespace

// web/script.js:591-612 (camera basis), :880-896, :899-929 (packing), :136-154 (TRS sliders)
extern "C" void orc_make_uniforms(const double cam_pos[3], double yaw, double pitch, const double sl[9],
                                  uint32_t width, uint32_t height, float time_value, float u[44]) {
  const double PI = 3.141592653589793;
  double cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
  double fwd[3] = {cp * sy, sp, -cp * cy};
  double tgt[3] = {cam_pos[0] + fwd[0], cam_pos[1] + fwd[1], cam_pos[2] + fwd[2]};
  double forward[3] = {tgt[0] - cam_pos[0], tgt[1] - cam_pos[1], tgt[2] - cam_pos[2]};
  normalize_js(forward);
  double up[3] = {0, 1, 0}, right[3], trueUp[3];
  cross_js(forward, up, right); normalize_js(right);
  cross_js(right, forward, trueUp);
  double fov = PI / 3, aspect = (double)width / (double)height, fl = 1 / std::tan(fov / 2);
  double scale[3] = {sl[0], sl[1], sl[2]};
  double rot[3] = {(sl[3] * PI) / 180 - PI, (sl[4] * PI) / 180, (sl[5] * PI) / 180};  // :148-150
  double pos[3] = {sl[6], sl[7], sl[8]};
  float model[16], inv[16];
  make_trs(rot, scale, pos, model);
  invert4(inv, model);
  for (int i = 0; i < 44; ++i) u[i] = 0.0f;
  for (int i = 0; i < 3; ++i) u[i] = (float)cam_pos[i];
  u[4] = 5.0f; u[5] = -5.0f; u[6] = 5.0f; u[7] = 0.3f;                    // :904-907
  for (int i = 0; i < 3; ++i) {
    u[8 + i] = (float)(right[i] * fl * aspect);
    u[12 + i] = (float)(trueUp[i] * fl);
    u[16 + i] = (float)(forward[i] * fl);
  }
  for (int i = 0; i < 16; ++i) u[20 + i] = inv[i];
  u[36] = (float)width; u[37] = (float)height;
  u[38] = time_value;  // reference always writes 0 here (F7); exposed for tests
}

// ============================================================ RNG + transcendental kernels
// web/compute.wgsl:135-143
extern "C" uint32_t orc_hash_u32(uint32_t seed) {
  uint32_t x = seed;
  x = (x ^ 61u) ^ (x >> 16);
  x = x + (x << 3);
  x = x ^ (x >> 4);
  x = x * 0x27d4eb2du;
  x = x ^ (x >> 15);
  return x;
}
// web/compute.wgsl:144-149 ; f32->u32 saturates
extern "C" float orc_rand2(float sx, float sy) {
  uint32_t xi = sat_u32((sx * 43758.0f) + 24521.0f);
  uint32_t yi = sat_u32((sy * 199999.0f) + 1423.0f);
  uint32_t mix = xi ^ (yi >> 6) ^ (yi << 11);
  return (float)orc_hash_u32(mix) * (1.0f / 4294967296.0f);
}

// Numerics contract: WGSL leaves sin/cos/pow accuracy implementation-defined.
// Both the oracle and the HIP kernels evaluate the SAME plain-f32 kernels below (only
// + - * /, floor and integer bit ops, fixed order), so results are reproducible bit for bit.
// sin/cos: Cody–Waite reduction by pi/2 (3 constants) + Cephes single-precision minimax
// polynomials on [-pi/4, pi/4]; max error ~1 ulp for |theta| < 1e4.
extern "C" void orc_sincos(float theta, float* s_out, float* c_out) {
  float k = floorf(theta * 0.63661977236758134308f + 0.5f);
  float r = theta - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cpoly = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
                - 0.5f * z + 1.0f;
  int q = sat_i32(k) & 3;
  float s = (q & 1) ? cpoly : sp;
  float c = (q & 1) ? sp : cpoly;
  if (q == 1 || q == 2) c = -c;   // q=1: (c,-s) ; q=2: (-s,-c) ; q=3: (-c, s)
  if (q == 2 || q == 3) s = -s;
  *s_out = s; *c_out = c;
}
// pow(x,y) for the gamma curve: x <= 0 (or tiny) -> 0; else 2^(y*log2 x) with
// log via atanh series on m in [sqrt(.5),sqrt(2)) and exp via degree-6 Taylor on [-.5,.5]*ln2.
extern "C" float orc_pow(float x, float y) {
  if (!(x >= 1.17549435e-38f)) return 0.0f;       // 0, negatives, denormals, NaN
  if (x > 3.0e38f) x = 3.0e38f;
  uint32_t bits; std::memcpy(&bits, &x, 4);
  int e = (int)((bits >> 23) & 0xFF) - 127;
  uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
  float m; std::memcpy(&m, &mb, 4);
  if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
  float f = m - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float ln_m = ((((z * (1.0f / 9.0f) + (1.0f / 7.0f)) * z + 0.2f) * z + (1.0f / 3.0f)) * z + 1.0f) * (2.0f * s);
  float L = (float)e + ln_m * 1.44269504088896341f;
  float p = y * L;
  float n = floorf(p + 0.5f);
  if (n < -125.0f) return 0.0f;
  if (n > 127.0f) n = 127.0f;
  float u = (p - n) * 0.69314718055994531f;
  float ex = (((((u * (1.0f / 720.0f) + (1.0f / 120.0f)) * u + (1.0f / 24.0f)) * u + (1.0f / 6.0f)) * u + 0.5f) * u
              + 1.0f) * u + 1.0f;
  uint32_t sb = (uint32_t)((int)n + 127) << 23;
  float sc; std::memcpy(&sc, &sb, 4);
  return ex * sc;
}

// acos for the fixed hemisphere directions of shader.wgsl:337 — pinned like sin/cos/pow above: Cephes-style asin polynomial
// on [0, 0.5], the half-angle identity above it; plain f32 ops in a fixed order.  The argument is clamped to [-1, 1].
extern "C" float orc_acos(float x) {
  x = fminf(fmaxf(x, -1.0f), 1.0f);
  const float ax = fabsf(x);
  const bool big = ax > 0.5f;
  const float z = big ? 0.5f * (1.0f - ax) : ax * ax;
  const float a = big ? sqrtf(z) : ax;
  const float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
  if (x >= 0.0f) return big ? 2.0f * p : 1.57079632679489661923f - p;
  return big ? 3.14159265358979323846f - 2.0f * p : 1.57079632679489661923f + p;
}

// web/shader.wgsl:176-180 (variant 1) ; shader.wgsl:320-326 (variant 2)
extern "C" uint32_t orc_frag_hash(int variant, uint32_t x) {
  if (variant == 1) {
    uint32_t v = x * 747796405u + 2891336453u;
    v = (v >> ((v >> 28) + 4u)) ^ v;
    return v;
  }
  uint32_t v = x;
  v = ((v >> 16) ^ v) * 0x45d9f3bu;
  v = ((v >> 16) ^ v) * 0x45d9f3bu;
  v = (v >> 16) ^ v;
  return v;
}
// web/shader.wgsl:181-185 = shader.wgsl:327-331 : the seeds' BIT PATTERNS are hashed
extern "C" float orc_frag_rand2(int variant, float sx, float sy) {
  uint32_t xi, yi; std::memcpy(&xi, &sx, 4); std::memcpy(&yi, &sy, 4);
  return (float)orc_frag_hash(variant, xi ^ (yi << 16)) * (1.0f / 4294967296.0f);
}

// ============================================================ traversal + shading (hot path 2)
namespace {

struct Scene {
  const float* nodes; uint32_t n_nodes;
  const float* tri_pos; const float* tri_nor; uint32_t n_tri;
  int max_stack, gi_samples, gi_bounces, soft_samples, mode;
  int enc;   // node record encoding: 1 = v1 (reference), 2 = v2
  const orc_surface* surf;   // material/texture extension (NULL = the reference's constant albedo)
  int frag;                  // 0 = compute.wgsl; 1 = web/shader.wgsl, 2 = shader.wgsl (older fragment shaders)
};
uint32_t* g_visit_counts = nullptr;        // analysis only, single-threaded use: internal steps per node
struct Counters {
  uint64_t rays_primary = 0, rays_shadow = 0, rays_bounce = 0, box = 0, tri = 0, hitupd = 0, pixels = 0, drops = 0;
};

// web/compute.wgsl:36-44,68-88
struct CachedNode { V3 mn, mx; int child0, child1; bool is_leaf; int tri_start, tri_count; };
inline CachedNode fetch_node(const Scene& S, int idx) {
  const float* b = S.nodes + (size_t)(uint32_t)idx * 8;
  CachedNode n;
  n.mn = v3(b[0], b[1], b[2]); n.mx = v3(b[4], b[5], b[6]);
  if (S.enc == 2) {   // v2: the same decode on u32 bit-casts
    const uint32_t A = f_bits(b[3]), B = f_bits(b[7]);
    n.is_leaf = (B >> 31) != 0;
    if (n.is_leaf) { n.tri_start = (int)A; n.tri_count = (int)(B & 0x7FFFFFFFu); n.child0 = -1; n.child1 = -1; }
    else { n.child0 = (int)A; n.child1 = (int)B; n.tri_start = -1; n.tri_count = 0; }
    return n;
  }
  n.is_leaf = (b[7] < 0.0f);
  if (n.is_leaf) { n.tri_start = sat_i32(b[3]); n.tri_count = -sat_i32(b[7]); n.child0 = -1; n.child1 = -1; }
  else { n.child0 = sat_i32(b[3]); n.child1 = sat_i32(b[7]); n.tri_start = -1; n.tri_count = 0; }
  return n;
}
inline V3 tri_p(const Scene& S, uint32_t idx) { const float* p = S.tri_pos + (size_t)idx * 4; return v3(p[0], p[1], p[2]); }
inline V3 tri_n(const Scene& S, uint32_t idx) { const float* p = S.tri_nor + (size_t)idx * 4; return v3(p[0], p[1], p[2]); }

// web/compute.wgsl:91-104
inline void aabb(V3 ro, V3 inv, V3 mn, V3 mx, float& tmin_o, float& tmax_o) {
  V3 t0 = (mn - ro) * inv, t1 = (mx - ro) * inv;
  V3 a = vmin(t0, t1), b = vmax(t0, t1);
  tmin_o = fmaxf(fmaxf(a.x, a.y), a.z);
  tmax_o = fminf(fminf(b.x, b.y), b.z);
}
// web/compute.wgsl:107-132 ; returns hit flag, fills t,u,v
inline bool tri_intersect(V3 ro, V3 rd, V3 v0, V3 v1, V3 v2, float& t, float& u, float& v) {
  V3 e1 = v1 - v0, e2 = v2 - v0;
  V3 p = cross(rd, e2);
  float det = dot(e1, p);
  V3 s = ro - v0;
  float invDet = 1.0f / det;
  u = dot(s, p) * invDet;
  V3 q = cross(s, e1);
  v = dot(rd, q) * invDet;
  t = dot(e2, q) * invDet;
  return (det > 1e-10f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) & (t > 1e-10f);
}

// web/shader.wgsl:133-163 = shader.wgsl:137-167 : the older shaders' test returns early and has other thresholds
inline bool tri_intersect_early(V3 ro, V3 rd, V3 v0, V3 v1, V3 v2, float& t, float& u, float& v) {
  V3 e1 = v1 - v0, e2 = v2 - v0;
  V3 p = cross(rd, e2);
  float det = dot(e1, p);
  if (det < 1e-8f) return false;
  float invDet = 1.0f / det;
  V3 s = ro - v0;
  u = dot(s, p) * invDet;
  if (u < 0.0f || u > 1.0f) return false;
  V3 q = cross(s, e1);
  v = dot(rd, q) * invDet;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(e2, q) * invDet;
  if (t <= 1e-6f) return false;
  return true;
}

struct Hit { float t; V3 n; uint32_t tri; float u, v; };

// web/compute.wgsl:259-346 — literal transcription incl. the 12-entry stack and far-child drop (F9)
Hit traverse(const Scene& S, V3 ro, V3 rd, V3 inv, float maxDist, Counters& C) {
  int stack[64];
  const int MAX_STACK = S.max_stack;
  int sp = 0, cur = 0;
  float bestT = maxDist; V3 bestN = v3(0, 0, 0); uint32_t bestTri = 0xFFFFFFFFu; float bestU = 0.0f, bestV = 0.0f;
  C.box += 1;  // the root is the only box not first tested as somebody's child
  for (;;) {
    if (cur < 0 || sp >= MAX_STACK) break;
    CachedNode node = fetch_node(S, cur);
    float hx, hy; aabb(ro, inv, node.mn, node.mx, hx, hy);
    if (hy >= fmaxf(hx, 0.0f) && hx <= bestT) {
      if (node.is_leaf) {
        for (int j = 0; j < node.tri_count; ++j) {
          uint32_t ti = (uint32_t)(node.tri_start + j);
          V3 va = tri_p(S, ti * 3 + 0), vb = tri_p(S, ti * 3 + 1), vc = tri_p(S, ti * 3 + 2);
          C.tri += 1;
          if (dot(cross(vb - va, vc - va), rd) > 0.0f) continue;          // :282
          float t, u, v;
          bool hit = S.frag ? tri_intersect_early(ro, rd, va, vb, vc, t, u, v) : tri_intersect(ro, rd, va, vb, vc, t, u, v);
          if (hit && t < bestT) {                                          // :286 strict <
            bestT = t; bestTri = ti; bestU = u; bestV = v; C.hitupd += 1;
            V3 n0 = tri_n(S, ti * 3 + 0), n1 = tri_n(S, ti * 3 + 1), n2 = tri_n(S, ti * 3 + 2);
            if (S.frag) bestN = normalize((n0 * ((1.0f - u) - v) + n1 * u) + n2 * v);   // web/shader.wgsl:309, shader.wgsl:227 (no fma)
            else bestN = normalize(fma3(n2, v, fma3(n1, u, n0 * ((1.0f - u) - v))));  // :291
          }
        }
        if (sp > 0) { sp -= 1; cur = stack[sp]; } else break;
      } else {
        int c0 = node.child0, c1 = node.child1;
        if (g_visit_counts) g_visit_counts[cur] += 1;
        CachedNode n0 = fetch_node(S, c0), n1 = fetch_node(S, c1);
        float b0x, b0y, b1x, b1y;
        aabb(ro, inv, n0.mn, n0.mx, b0x, b0y);
        aabb(ro, inv, n1.mn, n1.mx, b1x, b1y);
        C.box += 2;
        bool hit0 = (b0y >= fmaxf(b0x, 0.0f) && b0x <= bestT);
        bool hit1 = (b1y >= fmaxf(b1x, 0.0f) && b1x <= bestT);
        if (hit0 && hit1) {
          if (b0x <= b1x) {
            if (sp < MAX_STACK - 1) { stack[sp] = c1; sp += 1; } else C.drops += 1;
            cur = c0;
          } else {
            if (sp < MAX_STACK - 1) { stack[sp] = c0; sp += 1; } else C.drops += 1;
            cur = c1;
          }
        } else if (hit0) { cur = c0; }
        else if (hit1) { cur = c1; }
        else { if (sp > 0) { sp -= 1; cur = stack[sp]; } else break; }
      }
    } else {
      if (sp > 0) { sp -= 1; cur = stack[sp]; } else break;
    }
  }
  return Hit{bestT, bestN, bestTri, bestU, bestV};
}

// web/compute.wgsl:166-194 — walks only the child0 chain (F8)
bool shadow_ray(const Scene& S, V3 ro, V3 rd, float maxDist, Counters& C) {
  V3 inv = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
  int cur = 0;
  for (;;) {
    if (cur < 0) break;
    CachedNode node = fetch_node(S, cur);
    float hx, hy; aabb(ro, inv, node.mn, node.mx, hx, hy);
    C.box += 1;
    if ((hy >= fmaxf(hx, 0.0f)) & (hx <= maxDist)) {
      if (node.is_leaf) {
        for (int j = 0; j < node.tri_count; ++j) {
          uint32_t ti = (uint32_t)(node.tri_start + j);
          V3 va = tri_p(S, ti * 3 + 0), vb = tri_p(S, ti * 3 + 1), vc = tri_p(S, ti * 3 + 2);
          C.tri += 1;
          float t, u, v;
          bool hit = S.frag ? tri_intersect_early(ro, rd, va, vb, vc, t, u, v) : tri_intersect(ro, rd, va, vb, vc, t, u, v);
          if (hit && t < maxDist) return true;
        }
        break;
      } else cur = node.child0;
    } else break;
  }
  return false;
}

const float TWO_PI = 6.283185307179586f;

// web/compute.wgsl:153-163
V3 sample_hemisphere(V3 n, float s1, float s2) {
  float theta = TWO_PI * s1;
  float r = sqrtf(s2);
  float sn, cs; orc_sincos(theta, &sn, &cs);
  float x = r * cs, y = r * sn;
  float z = sqrtf(fmaxf(0.0f, 1.0f - s2));
  V3 up = (fabsf(n.y) < 0.999f) ? v3(0, 1, 0) : v3(1, 0, 0);
  V3 T = normalize(cross(up, n));
  V3 B = cross(n, T);
  return normalize((T * x + B * y) + n * z);
}

// web/compute.wgsl:197-256 ; time_h = f32(hash_u32(u32(uniforms.time)))
float direct_radiance(const Scene& S, V3 hit_os, V3 n_os, V3 light_os, uint32_t seed_id, float time_h, Counters& C) {
  V3 hitToLight = light_os - hit_os;
  V3 L = normalize(hitToLight);
  float diff = fmaxf(dot(n_os, L), 0.0f);
  if (diff <= 0.0f) return 0.0f;
  if (S.mode == ORC_MODE_PRIMARY) return diff;   // cfg 1-2: N.L only, no shadow rays
  uint32_t occ = 0;
  V3 up = (fabsf(n_os.y) > 0.99f) ? v3(1, 0, 0) : v3(0, 1, 0);
  V3 tangent = normalize(cross(up, n_os));
  V3 bitangent = cross(n_os, tangent);
  V3 ro2 = fma3(n_os, 0.001f, hit_os);
  float lightDist = length(hitToLight);
  const uint32_t GI = (uint32_t)S.gi_samples, SOFT = (uint32_t)S.soft_samples;
  for (uint32_t i = 0; i < SOFT; ++i) {
    float sqrtS = ceilf(sqrtf((float)GI));
    uint32_t cols = sat_u32(sqrtS);
    uint32_t rows = (GI + cols - 1u) / cols;
    uint32_t sx = i % cols, sy = i / cols;
    float jitter1 = orc_rand2((float)(seed_id + 17u) + time_h, (float)i + 0.37f);
    float jitter2 = orc_rand2((float)(seed_id + 91u) + time_h, (float)i + 1.13f);
    float s1 = ((float)sx + jitter1) / (float)cols;
    float s2 = ((float)sy + jitter2) / (float)rows;
    float theta = TWO_PI * s1;
    float r = sqrtf(s2);
    float sn, cs; orc_sincos(theta, &sn, &cs);
    float x = r * cs, y = r * sn;
    float z = sqrtf(fmaxf(0.0f, 1.0f - s2));
    V3 rd2 = normalize((tangent * x + bitangent * y) + n_os * z);
    bool visible = dot(rd2, normalize(hitToLight)) > 0.5f;
    C.rays_shadow += 1;
    bool hit = shadow_ray(S, ro2, rd2, lightDist - 0.001f, C);
    occ += (visible & hit) ? 1u : 0u;
  }
  float shadowFactor = fmaf(-(float)occ, 1.0f / (float)SOFT, 1.0f);
  return diff * shadowFactor;
}

inline V3 mat_mul4(const float* m, float x, float y, float z, float w) {  // columns * components, left to right
  return v3(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w,
            ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
            ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w);
}
inline V3 mat_mul3(const float* m, V3 v) {
  return v3((m[0] * v.x + m[4] * v.y) + m[8] * v.z,
            (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
            (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}

// Material / texture extension (north_star "simple material/texture shading"; the reference has none: compute.wgsl:372
// is a constant).  Definition, shared word for word with rt_trace.hip:hit_albedo and include/rtcore.h:
//   material id < 0 (or no surface bound)  -> the reference constant (1, 0.84, 0)
//   albedo = Kd, times the map_Kd texel when the material has a texture
//   uv = fma(uv2, v, fma(uv1, u, uv0 * ((1 - u) - v)))           (the form of the normal interpolation, compute.wgsl:291)
//   repeat wrap f = uv - floor(uv); x = min(u32(f.x * W), W-1); y = min(u32((1 - f.y) * H), H-1)   (OBJ v = 0 is the bottom row)
//   nearest texel, channel value = f32(byte) / 255
V3 hit_albedo(const Scene& S, const Hit& h) {
  const V3 ref = v3(1.0f, 0.84f, 0.0f);
  const orc_surface* sf = S.surf;
  if (!sf) return ref;
  const float* r = sf->tri_uv + 8 * (size_t)h.tri;
  const int32_t mat = (int32_t)f_bits(r[6]);
  if (mat < 0 || (uint32_t)mat >= sf->n_materials) return ref;
  const float* m = sf->materials + 4 * (size_t)mat;
  V3 kd = v3(m[0], m[1], m[2]);
  const uint32_t slot = f_bits(m[3]);
  if (slot >= sf->n_textures) return kd;
  const float w0 = (1.0f - h.u) - h.v;
  const float tu = fmaf(r[4], h.v, fmaf(r[2], h.u, r[0] * w0));
  const float tv = fmaf(r[5], h.v, fmaf(r[3], h.u, r[1] * w0));
  const float fu = tu - floorf(tu), fv = tv - floorf(tv);
  const uint32_t* tt = sf->tex_table + 4 * (size_t)slot;
  const uint32_t W = tt[1], H = tt[2];
  const uint32_t x = std::min(sat_u32(fu * (float)W), W - 1u), y = std::min(sat_u32((1.0f - fv) * (float)H), H - 1u);
  const uint32_t px = sf->texels[(size_t)tt[0] + (size_t)y * W + x];
  const V3 c = v3((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f);
  return kd * c;
}

struct Pixel { float r, g, b, a; uint32_t tri; float t; };
thread_local uint32_t* g_work_img = nullptr;   // optional per-pixel box-test count (analysis only)

// web/compute.wgsl:349-410
Pixel compute_lighting(const Scene& S, const float* U, V3 ro_w, V3 rd_w, uint32_t pixelIndex, float time, Counters& C) {
  const float* model = U + 20;
  V3 ro_os = mat_mul4(model, ro_w.x, ro_w.y, ro_w.z, 1.0f);
  V3 rd_os = normalize(mat_mul3(model, rd_w));
  V3 invRd = v3(1.0f / rd_os.x, 1.0f / rd_os.y, 1.0f / rd_os.z);
  V3 light_os = mat_mul4(model, U[4], U[5], U[6], U[7]);
  C.rays_primary += 1;
  Hit hit0 = traverse(S, ro_os, rd_os, invRd, 1e20f, C);
  if (hit0.t >= 1e19f) return Pixel{0, 0, 0, 1, 0xFFFFFFFFu, hit0.t};
  V3 hit_pos = fma3(rd_os, hit0.t, ro_os);
  V3 n_os = hit0.n;
  V3 albedo = hit_albedo(S, hit0);                      // compute.wgsl:372 when no surface is bound
  V3 indirect = v3(0, 0, 0);
  if (S.mode == ORC_MODE_PT) {
    const uint32_t GI = (uint32_t)S.gi_samples, NB = (uint32_t)S.gi_bounces;
    for (uint32_t i = 0; i < GI; ++i) {
      V3 throughput = albedo;
      V3 ro_b = fma3(n_os, 0.001f, hit_pos);
      V3 n_b = n_os;
      for (uint32_t b = 0; b < NB; ++b) {
        float bounceSeed = (float)b * 101.0f + (float)i;
        float pixelSeed = (float)pixelIndex / (U[36] * U[37]);
        float s1 = orc_rand2(bounceSeed, fmaf(pixelSeed, (float)i, 17.0f) + time);
        float s2 = orc_rand2(fmaf(pixelSeed, (float)i, 37.0f) + time, bounceSeed);
        V3 dir = sample_hemisphere(n_b, s1, s2);
        C.rays_bounce += 1;
        Hit hb = traverse(S, ro_b, dir, v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z), 1e20f, C);
        if (hb.t >= 1e19f) break;
        V3 pB = fma3(dir, hb.t, ro_b);
        V3 nB = hb.n;
        float dS = direct_radiance(S, pB, nB, light_os, pixelIndex, time, C);
        indirect = indirect + throughput * dS;
        throughput = throughput * (S.surf ? hit_albedo(S, hb) : albedo);   // :400 — the constant is every surface's albedo there
        ro_b = fma3(nB, 0.0001f, pB);
        n_b = nB;
      }
    }
    indirect = indirect / (float)GI;
  }
  float d0 = direct_radiance(S, hit_pos, n_os, light_os, pixelIndex, time, C);
  V3 col = fma3v(albedo, v3(d0, d0, d0), indirect);
  const float g = 1.0f / 2.2f;
  return Pixel{orc_pow(col.x, g), orc_pow(col.y, g), orc_pow(col.z, g), 1.0f, hit0.tri, hit0.t};
}

// ---------------------------------------------------------------- older fragment-pipeline shaders
// Both files share the traversal loop, the child0-chain shadow walk and this area-light estimate with compute.wgsl's
// predecessor constants; they differ from compute.wgsl in the triangle test (above), plain mul/add where compute.wgsl
// writes fma, the sky colour on a miss, the RNG and the GI estimator.

// web/shader.wgsl:27-36 = shader.wgsl:33-42
const float POISSON_DISK[8][2] = {{-0.7071f, 0.7071f}, {0.7071f, 0.7071f}, {0.7071f, -0.7071f}, {-0.7071f, -0.7071f},
                                  {-0.3536f, 0.9239f}, {0.9239f, 0.3536f}, {0.3536f, -0.9239f}, {-0.9239f, -0.3536f}};

// web/shader.wgsl:240-270 = shader.wgsl:355-384
float direct_radiance_disk(const Scene& S, V3 hit_os, V3 n_os, V3 light_os, float radius, Counters& C) {
  V3 L = normalize(light_os - hit_os);
  float diff = fmaxf(dot(n_os, L), 0.0f);
  if (diff <= 0.0f) return 0.0f;
  uint32_t occ = 0;
  V3 perp1 = normalize(cross(n_os, L));
  V3 perp2 = cross(L, perp1);
  const uint32_t SOFT = (uint32_t)S.soft_samples;
  for (uint32_t i = 0; i < SOFT; ++i) {
    float ox = POISSON_DISK[i][0] * radius, oy = POISSON_DISK[i][1] * radius;
    V3 samplePos = (light_os + perp1 * ox) + perp2 * oy;
    V3 ro2 = hit_os + n_os * 0.001f;
    V3 d = samplePos - hit_os;
    V3 rd2 = normalize(d);
    float sd = length(d);
    C.rays_shadow += 1;
    if (shadow_ray(S, ro2, rd2, sd - 0.001f, C)) occ += 1;
  }
  float shadowFactor = 1.0f - (float)occ / (float)SOFT;
  return diff * shadowFactor;
}

// shader.wgsl:170-172
inline bool is_safe_vec(V3 v) {
  return length(v) < 1e19f && fabsf(v.x) <= 1e30f && fabsf(v.y) <= 1e30f && fabsf(v.z) <= 1e30f;
}

// shader.wgsl:45-50,334-352 : HEMI_SAMPLES[idx] = ((idx % 8) * 2 + 1) / 16, ((idx / 8) * 2 + 1) / 16 (exact in f32)
V3 sample_hemisphere_fixed(V3 n, uint32_t idx) {
  float sx = (float)((idx & 7u) * 2u + 1u) / 16.0f, sy = (float)((idx >> 3) * 2u + 1u) / 16.0f;
  float theta = sx * 6.28318530718f;
  float phi = orc_acos(sqrtf(sy));
  float sinP, cosP, sinT, cosT;
  orc_sincos(phi, &sinP, &cosP);
  orc_sincos(theta, &sinT, &cosT);
  V3 up = (fabsf(n.y) < 0.999f) ? v3(0, 1, 0) : v3(1, 0, 0);
  V3 T = normalize(cross(up, n));
  V3 B = cross(n, T);
  return normalize((T * (sinP * cosT) + n * cosP) + B * (sinP * sinT));
}

// web/shader.wgsl:369-433 (variant 1) ; shader.wgsl:387-447 (variant 2).  uv = the fragment's interpolated quad coordinate.
Pixel compute_lighting_frag(const Scene& S, const float* U, V3 ro_w, V3 rd_w, float uvx, float uvy, Counters& C) {
  const float* model = U + 20;
  V3 ro_os = mat_mul4(model, ro_w.x, ro_w.y, ro_w.z, 1.0f);
  V3 rd_os = normalize(mat_mul3(model, rd_w));
  V3 invRd = v3(1.0f / rd_os.x, 1.0f / rd_os.y, 1.0f / rd_os.z);
  V3 light_os = mat_mul4(model, U[4], U[5], U[6], U[7]);
  const float radius = U[7];
  C.rays_primary += 1;
  Hit hit0 = traverse(S, ro_os, rd_os, invRd, 1e20f, C);
  if (hit0.t >= 1e19f) return Pixel{0.6f, 0.8f, 1.0f, 1.0f, 0xFFFFFFFFu, hit0.t};          // sky
  V3 hit_pos = ro_os + rd_os * hit0.t;
  V3 n_os = hit0.n;
  const V3 albedo = v3(1.0f, 0.84f, 0.0f);
  const float g = 1.0f / 2.2f;
  const uint32_t GI = (uint32_t)S.gi_samples;
  if (S.frag == 1) {
    float directScalar = direct_radiance_disk(S, hit_pos, n_os, light_os, radius, C);
    V3 Lo = albedo * directScalar;
    V3 indirect = v3(0, 0, 0);
    const uint32_t NB = (uint32_t)S.gi_bounces;
    for (uint32_t i = 0; i < GI; ++i) {
      V3 throughput = albedo;
      V3 ro_b = hit_pos + n_os * 0.001f;
      V3 n_b = n_os;
      for (uint32_t b = 0; b < NB; ++b) {
        float rx = orc_frag_rand2(1, (float)i, (float)b + 17.0f);
        float ry = orc_frag_rand2(1, (float)i + 37.0f, (float)b);
        V3 dir = sample_hemisphere(n_b, rx, ry);                       // :189-204, the same statements as compute.wgsl:153-163
        C.rays_bounce += 1;
        Hit h1 = traverse(S, ro_b, dir, v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z), 50.0f, C);
        if (h1.t >= 50.0f) break;
        V3 p1 = ro_b + dir * h1.t;
        V3 nn1 = h1.n;
        float dS = direct_radiance_disk(S, p1, nn1, light_os, radius, C);
        indirect = indirect + throughput * dS;
        throughput = throughput * albedo;
        ro_b = p1 + nn1 * 0.001f;
        n_b = nn1;
      }
    }
    indirect = indirect / (float)GI;
    V3 col = Lo + indirect;
    return Pixel{orc_pow(col.x, g), orc_pow(col.y, g), orc_pow(col.z, g), 1.0f, hit0.tri, hit0.t};
  }
  if (!is_safe_vec(n_os)) n_os = v3(0, 1, 0);
  float direct = direct_radiance_disk(S, hit_pos, n_os, light_os, radius, C);
  float indirect = 0.0f;
  const float giProb = 0.9f;
  for (uint32_t i = 0; i < GI; ++i) {
    float sx = uvx + (float)i * 0.031f, sy = uvy + (float)i * 0.047f;
    if (orc_frag_rand2(2, sx, sy) > giProb) continue;
    V3 dir1 = sample_hemisphere_fixed(n_os, i);
    V3 ro1 = hit_pos + n_os * 0.001f;
    C.rays_bounce += 1;
    Hit h1 = traverse(S, ro1, dir1, v3(1.0f / dir1.x, 1.0f / dir1.y, 1.0f / dir1.z), 50.0f, C);
    if (h1.t < 50.0f) {
      V3 p1 = ro1 + dir1 * h1.t;
      V3 nn1 = h1.n;
      if (!is_safe_vec(nn1)) nn1 = v3(0, 1, 0);
      indirect = indirect + direct_radiance_disk(S, p1, nn1, light_os, radius, C) / giProb;
    }
  }
  indirect = indirect / (float)GI;
  float k = direct + indirect * 0.6f;
  V3 col = albedo * k;
  return Pixel{orc_pow(col.x, g), orc_pow(col.y, g), orc_pow(col.z, g), 1.0f, hit0.tri, hit0.t};
}

inline uint8_t unorm8(float c) {  // rgba8unorm store: round(clamp(c,0,1)*255), NaN -> 0
  float v = fminf(fmaxf(c, 0.0f), 1.0f);
  return (uint8_t)sat_u32(floorf(v * 255.0f + 0.5f));
}

// web/compute.wgsl:413-446
void render_pixel(const Scene& S, const float* U, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h, float time_h,
                  uint8_t* rgba8, uint32_t* hit_tri, float* hit_t, Counters& C) {
  float uvx = ((float)gx + 0.5f) / (float)w, uvy = ((float)gy + 0.5f) / (float)h;
  V3 ro = v3(U[0], U[1], U[2]);
  float sx = fmaf(uvx, 2.0f, -1.0f), sy = fmaf(uvy, 2.0f, -1.0f);
  V3 camR = v3(U[8], U[9], U[10]), camU = v3(U[12], U[13], U[14]), camF = v3(U[16], U[17], U[18]);
  V3 rd = normalize(fma3(camR, sx, fma3(camU, sy, camF)));
  float px = fminf(fmaxf(uvx * (float)w, 0.0f), (float)(w - 1));
  float py = fminf(fmaxf(uvy * (float)h, 0.0f), (float)(h - 1));
  uint32_t ix = sat_u32(px), iy = sat_u32(py);
  uint32_t pixelIndex = iy * w + ix;
  const uint64_t box0 = C.box;
  Pixel p;
  if (S.frag) {
    // vs_main / fs_main (web/shader.wgsl:61-67,165-173,435-440): the quad's uv at the fragment centre; framebuffer row 0 is
    // the TOP row and clip-space y points up, so uv.y = 1 - (gy + 0.5) / h.  (A rasteriser's interpolation of uv is
    // implementation-defined in its last bits; this is the definition both sides of the parity test use.)
    const float fuy = 1.0f - uvy;
    const float fsx = uvx * 2.0f - 1.0f, fsy = fuy * 2.0f - 1.0f;
    V3 frd = normalize((camR * fsx + camU * fsy) + camF);
    p = compute_lighting_frag(S, U, ro, frd, uvx, fuy, C);
  } else p = compute_lighting(S, U, ro, rd, pixelIndex, time_h, C);
  size_t o = (size_t)gy * w + gx;
  if (g_work_img) g_work_img[o] = (uint32_t)(C.box - box0);
  if (rgba8) { rgba8[4 * o + 0] = unorm8(p.r); rgba8[4 * o + 1] = unorm8(p.g); rgba8[4 * o + 2] = unorm8(p.b); rgba8[4 * o + 3] = unorm8(p.a); }
  if (hit_tri) hit_tri[o] = p.tri;
  if (hit_t) hit_t[o] = p.t;
  C.pixels += 1;
}

}  // namespace

static uint32_t* g_work_request = nullptr;
static const orc_surface* g_surface = nullptr;
extern "C" void orc_set_surface(const orc_surface* s) { g_surface = s; }

// BVH-order surface records from the loadObj-style arrays: record i describes input triangle order[i]
extern "C" void orc_gather_tri_uv(const float* texcoords, const uint32_t* tex_idx, const int32_t* material_ids,
                                  const uint32_t* order, uint32_t n_tri, float* tri_uv) {
  for (uint32_t i = 0; i < n_tri; ++i) {
    const uint32_t t = order[i];
    float* r = tri_uv + 8 * (size_t)i;
    for (int k = 0; k < 3; ++k) {
      const size_t ti = tex_idx[3 * (size_t)t + k];
      r[2 * k + 0] = texcoords[2 * ti + 0];
      r[2 * k + 1] = texcoords[2 * ti + 1];
    }
    r[6] = bits_f((uint32_t)(material_ids ? material_ids[t] : -1));
    r[7] = 0.0f;
  }
}

extern "C" void orc_set_work_image(uint32_t* img) { g_work_request = img; }
extern "C" void orc_set_visit_counts(uint32_t* per_node) { g_visit_counts = per_node; }   // closest-hit internal steps per node (run with n_threads = 1)   // box tests per pixel of the next orc_render calls

extern "C" int orc_render(const float* nodes, uint32_t n_nodes, const float* tri_pos, const float* tri_nor,
                          uint32_t n_tri, const float U[44], const orc_render_opts* o,
                          uint8_t* rgba8, uint32_t* hit_tri, float* hit_t, orc_render_stats* stats) {
  if (!nodes || !tri_pos || !tri_nor || !U || !o) return -1;
  if (o->max_stack < 1 || o->max_stack > 64 || o->row_step < 1) return -2;
  if (n_nodes == 0 || n_tri == 0) return -3;
  Scene S{nodes, n_nodes, tri_pos, tri_nor, n_tri, o->max_stack, o->gi_samples, o->gi_bounces, o->soft_samples, o->mode,
          o->node_encoding == 2 ? 2 : 1, g_surface,
          o->mode == ORC_MODE_FRAG_WEB ? 1 : (o->mode == ORC_MODE_FRAG_ROOT ? 2 : 0)};
  if (S.frag && (o->soft_samples < 1 || o->soft_samples > 8)) return -4;     // POISSON_DISK has 8 entries
  if (S.frag == 2 && (o->gi_samples < 1 || o->gi_samples > 16)) return -4;   // HEMI_SAMPLES has 16
  if (S.frag) S.surf = nullptr;                                               // constant albedo in both files
  const uint32_t w = sat_u32(U[36]), h = sat_u32(U[37]);
  const float time_h = (float)orc_hash_u32(sat_u32(U[38]));      // compute.wgsl:444
  int r0 = std::max(0, o->row_begin), r1 = std::min<int>((int)h, o->row_end), rs = o->row_step;
  int nthr = std::max(1, o->n_threads);
  std::vector<Counters> cs(nthr);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&](int tid) {
    Counters& C = cs[tid];
    g_work_img = g_work_request;
    int k = 0;
    for (int y = r0; y < r1; y += rs, ++k) {
      if (k % nthr != tid) continue;
      for (uint32_t x = 0; x < w; ++x) render_pixel(S, U, x, (uint32_t)y, w, h, time_h, rgba8, hit_tri, hit_t, C);
    }
  };
  if (nthr == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (auto& C : cs) {
      stats->rays_primary += C.rays_primary; stats->rays_shadow += C.rays_shadow; stats->rays_bounce += C.rays_bounce;
      stats->box_tests += C.box; stats->tri_tests += C.tri; stats->hit_updates += C.hitupd;
      stats->pixels += C.pixels; stats->stack_drops += C.drops;
    }
    stats->seconds = sec;
  }
  return 0;
}
