// rt_device.h — gfx950 device-side numerics shared by the build and traversal kernels.
//
// Float rules (DESIGN.md "numerics contract"): strict IEEE f32 — compiled with
// -ffp-contract=off, correctly rounded divide/sqrt (hipcc default), denormals kept; fma only
// where the reference WGSL writes fma; sums left to right; fminf/fmaxf (v_min/v_max_f32
// return the non-NaN operand); float->int conversions saturate like WGSL.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

struct V3 { float x, y, z; };
RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_DEV float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV float length3(V3 a) { return sqrtf(dot3(a, a)); }
RT_DEV V3 normalize3(V3 a) { return a / length3(a); }
RT_DEV V3 fma3(V3 a, float s, V3 c) { return v3(__fmaf_rn(a.x, s, c.x), __fmaf_rn(a.y, s, c.y), __fmaf_rn(a.z, s, c.z)); }
RT_DEV V3 rcp3(V3 a) { return v3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
RT_DEV V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }

// WGSL f32->u32 / i32: saturating, NaN -> 0 (v_cvt_u32_f32 / v_cvt_i32_f32 do exactly this,
// but spell it out so the semantics do not hang on an instruction selection).
RT_DEV uint32_t sat_u32(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
RT_DEV int32_t sat_i32(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)f;
}

// Slots A / B of the 32-byte node record (rt_internal.h): v1 = indices as f32 VALUES, leaf B = -count
// (web/script.js:546-559, exact below 2^24, F11); v2 = u32 bit-casts in the same slots, leaf flag =
// bit 31 of B, low 31 bits = count.  The floats are only ever moved, never computed on.
RT_DEV float enc_index(bool v2, uint32_t i) { return v2 ? __uint_as_float(i) : (float)i; }
RT_DEV float enc_leaf_b(bool v2, uint32_t count) { return v2 ? __uint_as_float(0x80000000u | count) : -(float)count; }
RT_DEV bool dec_is_leaf(bool v2, float b) { return v2 ? ((__float_as_uint(b) >> 31) != 0u) : (b < 0.0f); }
RT_DEV int32_t dec_index(bool v2, float a) { return v2 ? (int32_t)__float_as_uint(a) : sat_i32(a); }
RT_DEV int32_t dec_count(bool v2, float b) { return v2 ? (int32_t)(__float_as_uint(b) & 0x7FFFFFFFu) : -sat_i32(b); }

// web/morton_compute.wgsl:18-31
RT_DEV uint32_t expand_bits(uint32_t v) {
  uint32_t x = v & 0x3FFu;
  x = (x * 0x00010001u) & 0xFF0000FFu;
  x = (x * 0x00000101u) & 0x0F00F00Fu;
  x = (x * 0x00000011u) & 0xC30C30C3u;
  x = (x * 0x00000005u) & 0x49249249u;
  return x;
}
RT_DEV uint32_t quantize10(float x) { return sat_u32(fminf(fmaxf(floorf(x * 1024.0f), 0.0f), 1023.0f)); }

// web/compute.wgsl:135-149
RT_DEV uint32_t hash_u32(uint32_t x) {
  x = (x ^ 61u) ^ (x >> 16);
  x = x + (x << 3);
  x = x ^ (x >> 4);
  x = x * 0x27d4eb2du;
  x = x ^ (x >> 15);
  return x;
}
RT_DEV float rand2(float sx, float sy) {
  uint32_t xi = sat_u32((sx * 43758.0f) + 24521.0f);
  uint32_t yi = sat_u32((sy * 199999.0f) + 1423.0f);
  uint32_t mix = xi ^ (yi >> 6) ^ (yi << 11);
  return (float)hash_u32(mix) * (1.0f / 4294967296.0f);
}

// Pinned transcendental kernels (WGSL leaves sin/cos/pow accuracy implementation-defined):
// Cody–Waite pi/2 reduction + Cephes f32 minimax polynomials; plain f32 ops in a fixed order.
RT_DEV void rt_sincos(float theta, float* s_out, float* c_out) {
  float k = floorf(theta * 0.63661977236758134308f + 0.5f);
  float r = theta - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  int q = sat_i32(k) & 3;
  float s = (q & 1) ? cp : sp;
  float c = (q & 1) ? sp : cp;
  if (q == 1 || q == 2) c = -c;
  if (q == 2 || q == 3) s = -s;
  *s_out = s; *c_out = c;
}
RT_DEV float rt_pow(float x, float y) {
  if (!(x >= 1.17549435e-38f)) return 0.0f;
  if (x > 3.0e38f) x = 3.0e38f;
  uint32_t bits = __float_as_uint(x);
  int e = (int)((bits >> 23) & 0xFF) - 127;
  float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
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
  float ex = (((((u * (1.0f / 720.0f) + (1.0f / 120.0f)) * u + (1.0f / 24.0f)) * u + (1.0f / 6.0f)) * u + 0.5f) * u + 1.0f) * u + 1.0f;
  float sc = __uint_as_float((uint32_t)((int)n + 127) << 23);
  return ex * sc;
}

// acos for the fixed hemisphere directions of shader.wgsl:337, pinned like sin/cos/pow: asin polynomial on [0, 0.5] and the
// half-angle identity above it, plain f32 ops in a fixed order; argument clamped to [-1, 1].
RT_DEV float rt_acos(float x) {
  x = fminf(fmaxf(x, -1.0f), 1.0f);
  const float ax = fabsf(x);
  const bool big = ax > 0.5f;
  const float z = big ? 0.5f * (1.0f - ax) : ax * ax;
  const float a = big ? sqrtf(z) : ax;
  const float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
  if (x >= 0.0f) return big ? 2.0f * p : 1.57079632679489661923f - p;
  return big ? 3.14159265358979323846f - 2.0f * p : 1.57079632679489661923f + p;
}

// RNGs of the older fragment shaders: web/shader.wgsl:176-185 (V = 1), shader.wgsl:320-331 (V = 2); the seeds' bit patterns are hashed
template <int V>
RT_DEV uint32_t frag_hash(uint32_t x) {
  if (V == 1) {
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
template <int V>
RT_DEV float frag_rand2(float sx, float sy) {
  return (float)frag_hash<V>(__float_as_uint(sx) ^ (__float_as_uint(sy) << 16)) * (1.0f / 4294967296.0f);
}

// order-preserving float <-> uint map (total order, -0 < +0) for min/max reductions/atomics
RT_DEV uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
RT_DEV float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
