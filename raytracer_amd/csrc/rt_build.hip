// rt_build.hip — hot path 1: Morton LBVH build on gfx950 (wave64).
//
// Pipeline (the reference only has the Morton shader,
// web/morton_compute.wgsl, and never dispatches it — F4):
//   k_bounds*      centroid min/max reduction (order-independent => bit-exact)
//   k_morton       morton_compute.wgsl:33-69, one thread per triangle
//   k_onesweep     onesweep LSD radix sort, 3 x 10-bit digits, (key, payload) pairs, STABLE:
//                  global digit histograms from k_morton, then per digit a single pass with
//                  grouped decoupled look-back, wave64 ballot/popcount ranking (10 ballots per key) and an
//                  LDS-staged scatter
//   (Karras 2012 radix tree, ties broken by index: inside k_refit_local / k_refit_mid, no kernel or arrays of its own)
//   k_refit_*      one thread per leaf: emit BVH-ordered triangle buffers + leaf record, then climb the nodes inside
//                  the thread's 256-leaf block through LDS; nodes above the blocks are range queries over leaf boxes
//                  and sparse tables of block boxes (no hand-off between workgroups inside a launch)
//   k_pack_*       derive the traversal layout (64-byte two-child records, hoisted edges)
// All kernels are HBM-bound integer/min-max work: coalesced 4-16 B/lane accesses, LDS staging,
// no MFMA.
#include "rt_device.h"
#include "rt_internal.h"

namespace {

constexpr int BLK = 256;
constexpr int RADIX_BITS = 10;                     // 30-bit Morton keys = 3 digits of 10 bits: three onesweep passes
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_PASSES = 3;
constexpr int DPT = RADIX / BLK;                   // digits owned by a thread: 4 consecutive ones
#ifndef RT_SORT_KPT
#define RT_SORT_KPT 20                             // 16 / 20 / 24 keys per thread: sort 92 / 86 / 88 us at 1.31 M keys, 502 / 362 / 347 us at 10.5 M
#endif
constexpr int SORT_KPT = RT_SORT_KPT;              // keys per thread
constexpr int SORT_TILE = BLK * SORT_KPT;          // 5120 keys per tile
constexpr uint32_t ST_AGG = 1u << 30, ST_INCL = 2u << 30, ST_MASK = (1u << 30) - 1u;
constexpr uint32_t SPIN_LIMIT = 1u << 24;          // bounded look-back spin (error flag instead of a hang)

// look-back words of one sort pass: n_tiles rows of 2 KB (u16 per digit), then, 4 KB-aligned, one 4 KB row (u32 per digit) per group
__host__ __device__ inline uint32_t sort_group_rows_offset(uint32_t n_tiles) { return (n_tiles * (uint32_t)(RADIX * 2) + 4095u) & ~4095u; }

inline int grid_for(uint32_t n, int per_block, int cap = 2048) {
  long g = ((long)n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

RT_DEV V3 load_vtx(const float* pos, uint32_t i) { return v3(pos[3 * (size_t)i + 0], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]); }
// web/morton_compute.wgsl:52 — (p0 + p1 + p2) / 3.0, left to right, true division
RT_DEV V3 tri_centroid(const float* pos, const uint32_t* vidx, uint32_t t) {
  V3 p0 = load_vtx(pos, vidx[3 * (size_t)t + 0]), p1 = load_vtx(pos, vidx[3 * (size_t)t + 1]), p2 = load_vtx(pos, vidx[3 * (size_t)t + 2]);
  return ((p0 + p1) + p2) / 3.0f;
}

RT_DEV uint32_t wave_min_u32(uint32_t v) {
  for (int o = 32; o >= 1; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
RT_DEV uint32_t wave_max_u32(uint32_t v) {
  for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}

// ---------------------------------------------------------------- mesh layout in HBM
// The loadObj arrays arrive indexed: vertex / normal / texcoord triples into three attribute arrays, material ids.  The refit
// reads a triangle's attributes THROUGH THE SORT ORDER: from the arrays as they come that is nine scattered 8- or 12-byte
// reads per triangle, each pulling a 64-byte sector (round 2: 884 MB of fabric traffic for 524 MB of algorithmic bytes, 1.8 x,
// with one 48-byte index record per triangle in between).  So the mesh is de-indexed ONCE, at upload (rt_set_mesh /
// rt_set_surface), into one 128-byte record per input triangle - the build then reads exactly two sectors per triangle:
//   {p0.xyz p1.x | p1.yz p2.xy | p2.z n0.xyz | n1.xyz n2.x | n2.yz uv0 | uv1 uv2 | bits(material) - - - | - - - -}
// (uv / material slots are filled by rt_set_surface).  Copies only: every float keeps its bits.
constexpr int MESH_REC = 8;       // float4 per record
__global__ __launch_bounds__(BLK) void k_pack_mesh(const float* __restrict__ pos, const float* __restrict__ nor, const uint32_t* __restrict__ vidx,
                                                   const uint32_t* __restrict__ nidx, const float* __restrict__ texcoords,
                                                   const uint32_t* __restrict__ tex_idx, const int32_t* __restrict__ material_ids,
                                                   uint32_t n_tri, float4* __restrict__ mesh_rec, float* __restrict__ tri_in_pos /* [9T] or NULL */) {
  const uint32_t t = blockIdx.x * BLK + threadIdx.x;
  if (t >= n_tri) return;
  const V3 p0 = load_vtx(pos, vidx[3 * (size_t)t + 0]), p1 = load_vtx(pos, vidx[3 * (size_t)t + 1]), p2 = load_vtx(pos, vidx[3 * (size_t)t + 2]);
  const V3 n0 = load_vtx(nor, nidx[3 * (size_t)t + 0]), n1 = load_vtx(nor, nidx[3 * (size_t)t + 1]), n2 = load_vtx(nor, nidx[3 * (size_t)t + 2]);
  float4* rec = mesh_rec + (size_t)t * MESH_REC;
  float2 a = make_float2(0.0f, 0.0f), b = a, c = a;
  uint32_t mat = 0xFFFFFFFFu;
  if (tex_idx) {
    const float2* tc = reinterpret_cast<const float2*>(texcoords);
    a = tc[tex_idx[3 * (size_t)t + 0]]; b = tc[tex_idx[3 * (size_t)t + 1]]; c = tc[tex_idx[3 * (size_t)t + 2]];
    if (material_ids) mat = (uint32_t)material_ids[t];
  }
  rec[0] = make_float4(p0.x, p0.y, p0.z, p1.x);
  rec[1] = make_float4(p1.y, p1.z, p2.x, p2.y);
  rec[2] = make_float4(p2.z, n0.x, n0.y, n0.z);
  rec[3] = make_float4(n1.x, n1.y, n1.z, n2.x);
  rec[4] = make_float4(n2.y, n2.z, a.x, a.y);
  rec[5] = make_float4(b.x, b.y, c.x, c.y);
  rec[6] = make_float4(__uint_as_float(mat), 0.0f, 0.0f, 0.0f);
  if (tri_in_pos) {                                      // the positions once more, 36 bytes per triangle in input order: k_bounds streams them
    float* q = tri_in_pos + 9 * (size_t)t;
    q[0] = p0.x; q[1] = p0.y; q[2] = p0.z; q[3] = p1.x; q[4] = p1.y; q[5] = p1.z; q[6] = p2.x; q[7] = p2.y; q[8] = p2.z;
  }
}

// ---------------------------------------------------------------- centroid bounds
// No same-address atomics: every block leaves its six ordered-uint partials in its own slot (2048 blocks x 6 same-address
// atomicMin/Max serialised at ~26 ns each: 53 of the kernel's 62 us at 1.3 M triangles); every block of k_morton folds the
// slots for itself (30 KB of L2 hits; as a launch of its own the fold cost 4 us + a launch gap.  A "last block folds"
// ticket inside k_bounds needs an agent-scope release fence per block, and on this chip that is a write-back of the XCD's L2:
// 1280 of them took 0.22 ms).
// Also keeps every centroid (12 B/triangle, coalesced) so that k_morton streams them instead of repeating the
// three-vertex gather through the index buffer (48 B/triangle of scattered reads).
// Reads the de-indexed positions k_pack_mesh left at upload (36 bytes per triangle, streamed): through the index buffer every
// vertex was a scattered 12-byte read that pulled a whole line - 30 us at 1.3 M triangles, and 4.4 ms at 99.6 M where a bare
// stream of the same bytes takes 1.2 ms (tools/ubench_bounds).
constexpr int BOUNDS_MAX_BLOCKS = 2048;
__global__ __launch_bounds__(BLK) void k_bounds(const float* __restrict__ tri_in_pos /* [9T], input order */,
                                                uint32_t n_tri, uint32_t* part /* [6][gridDim.x] */,
                                                float* __restrict__ centroids /* [3T] */,
                                                uint32_t* __restrict__ hist, uint32_t* __restrict__ tile_counter) {
  if (blockIdx.x == 0) {
    // the sort's global digit histograms and tile tickets start at zero (saves three memset launches)
    for (int i = threadIdx.x; i < SORT_PASSES * RADIX; i += BLK) hist[i] = 0u;
    if (threadIdx.x < 4) tile_counter[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < 64 * 32; i += BLK) tile_counter[32 + i] = 0u;      // the refit's lists of block-crossing nodes (CROSS_SHARDS x CROSS_STRIDE)
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  constexpr int BU = 2;                                  // triangles per thread and step, their loads issued together
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t t0 = (size_t)blockIdx.x * BLK + threadIdx.x; t0 < n_tri; t0 += BU * stride) {
    V3 p[BU][3];
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const size_t t = t0 + u * stride;
      if (t < n_tri) { p[u][0] = load_vtx(tri_in_pos, 3 * t + 0); p[u][1] = load_vtx(tri_in_pos, 3 * t + 1); p[u][2] = load_vtx(tri_in_pos, 3 * t + 2); }
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const size_t t = t0 + u * stride;
      if (t < n_tri) {
        const V3 c = ((p[u][0] + p[u][1]) + p[u][2]) / 3.0f;             // web/morton_compute.wgsl:52, as tri_centroid
        centroids[3 * t + 0] = c.x; centroids[3 * t + 1] = c.y; centroids[3 * t + 2] = c.z;
        const uint32_t e[3] = {f2ord(c.x), f2ord(c.y), f2ord(c.z)};
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], e[k]); mx[k] = max(mx[k], e[k]); }
      }
    }
  }
  __shared__ uint32_t s[6][BLK / 64];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint32_t a = wave_min_u32(mn[k]), b = wave_max_u32(mx[k]);
    if (lane == 0) { s[k][wave] = a; s[3 + k][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    uint32_t v = s[k][0];
    for (int w = 1; w < BLK / 64; ++w) v = (k < 3) ? min(v, s[k][w]) : max(v, s[k][w]);
    part[(size_t)k * gridDim.x + blockIdx.x] = v;        // component-major: the fold reads them coalesced
  }}

// fold of k_bounds' block partials by one block (wave w: components w and w + 4), then the `Bounds` uniform of
// morton_compute.wgsl:3-6: invExtent = 1/(max-min), 0 where the extent is 0.  s6: six words of LDS; all threads of the block call.
RT_DEV void fold_bounds(const uint32_t* __restrict__ part, uint32_t n_blocks, uint32_t* s6, V3& bmin, V3& inv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < 6; k += BLK / 64) {
    uint32_t v = (k < 3) ? 0xFFFFFFFFu : 0u;
    uint32_t x[BOUNDS_MAX_BLOCKS / 64];                  // all loads of a wave issued before the first is used
#pragma unroll
    for (int it = 0; it < BOUNDS_MAX_BLOCKS / 64; ++it) {
      const uint32_t bi = (uint32_t)lane + 64u * it;
      x[it] = (bi < n_blocks) ? part[(size_t)k * n_blocks + bi] : v;
    }
#pragma unroll
    for (int it = 0; it < BOUNDS_MAX_BLOCKS / 64; ++it) v = (k < 3) ? min(v, x[it]) : max(v, x[it]);
    v = (k < 3) ? wave_min_u32(v) : wave_max_u32(v);
    if (lane == 0) s6[k] = v;
  }
  __syncthreads();
  float lo[3], iv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = ord2f(s6[c]);
    const float ext = ord2f(s6[3 + c]) - lo[c];
    iv[c] = (ext > 0.0f) ? (1.0f / ext) : 0.0f;
  }
  bmin = v3(lo[0], lo[1], lo[2]); inv = v3(iv[0], iv[1], iv[2]);
}

// ---------------------------------------------------------------- Morton codes
// Also the sort's global digit histograms (the keys are in registers here: no separate histogram pass over them): one
// LDS histogram per block and digit position, flushed with one global atomic per non-empty bin.  Few, fat blocks: the flush
// costs one same-address atomic per block and bin.
__global__ __launch_bounds__(BLK) void k_morton(const float* __restrict__ centroids,
                                                uint32_t n_tri, const uint32_t* __restrict__ bounds_part, uint32_t n_bounds_blocks, float* __restrict__ bounds_f,
                                                uint32_t* __restrict__ codes, uint32_t* __restrict__ hist,
                                                uint32_t* __restrict__ tile_status, size_t n_status) {
  __shared__ uint32_t h[SORT_PASSES][RADIX];
  for (int i = threadIdx.x; i < SORT_PASSES * RADIX; i += BLK) (&h[0][0])[i] = 0u;
  // the look-back words of the sort passes start at zero (their own memset launch otherwise)
  for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n_status; i += (size_t)gridDim.x * BLK) tile_status[i] = 0u;
  __shared__ uint32_t s6[6];
  V3 bmin, inv;
  fold_bounds(bounds_part, n_bounds_blocks, s6, bmin, inv);          // (its barrier also covers the zeroed histograms)
  if (blockIdx.x == 0 && threadIdx.x == 0) {                        // the uniform itself, for rt_get_bounds
    bounds_f[0] = bmin.x; bounds_f[1] = bmin.y; bounds_f[2] = bmin.z; bounds_f[3] = 0.0f;
    bounds_f[4] = inv.x; bounds_f[5] = inv.y; bounds_f[6] = inv.z; bounds_f[7] = 0.0f;
  }
  // four triangles per step, their centroids loaded before the first is used (a thread walks ~16 triangles: one at a time
  // the loop is a chain of exposed load latencies)
  constexpr int MU = 4;
  const uint32_t stride = gridDim.x * BLK;
  for (uint32_t t0 = blockIdx.x * BLK + threadIdx.x; t0 < n_tri; t0 += MU * stride) {
    V3 ctr[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      const uint32_t t = t0 + u * stride;
      if (t < n_tri) ctr[u] = v3(centroids[3 * (size_t)t + 0], centroids[3 * (size_t)t + 1], centroids[3 * (size_t)t + 2]);   // :52, from k_bounds
    }
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      const uint32_t t = t0 + u * stride;
      if (t < n_tri) {
        V3 nrm = (ctr[u] - bmin) * inv;                                   // :55
        uint32_t code = (expand_bits(quantize10(nrm.x)) << 2) | (expand_bits(quantize10(nrm.y)) << 1) |
                        expand_bits(quantize10(nrm.z));                  // :58-68
        codes[t] = code;
#pragma unroll
        for (int p = 0; p < SORT_PASSES; ++p) atomicAdd(&h[p][(code >> (RADIX_BITS * p)) & (RADIX - 1)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SORT_PASSES * RADIX; i += BLK) {
    const uint32_t v = (&h[0][0])[i];
    if (v) atomicAdd(&hist[i], v);
  }
}

// ---------------------------------------------------------------- onesweep radix sort
// block-wide exclusive scan of one value per thread (BLK = 256 = 4 waves)
RT_DEV uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BLK / 64; ++w) { uint32_t x = s_wave[w]; if (w < wave) base += x; tot += x; }
  if (total) *total = tot;
  __syncthreads();
  return base + inc - v;
}

// One LSD pass over a 10-bit digit.  Thread t owns the four consecutive digits 4t .. 4t+3: their tile counts travel as four
// look-back words {flag | count} in one 16-byte write-through store, and the digit
// bases (exclusive scan of the pass's global histogram, which k_morton left as raw counts) are scanned here, per block,
// instead of by a launch of their own.  FIRST: the payload of key i is i itself, nothing to read.
#ifdef RT_SORT_TIMELINE   // analysis build only (make EXTRA=-DRT_SORT_TIMELINE): stage clocks of every onesweep tile (tools/sort_timeline.py)
__device__ unsigned long long g_sort_tl[SORT_PASSES][4096][8];
#define SORT_STAMP(k) do { if (tid == 0 && tile < 4096u) g_sort_tl[shift / RADIX_BITS][tile][k] = wall_clock64(); } while (0)
#else
#define SORT_STAMP(k) do { } while (0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool FIRST>
__global__ __launch_bounds__(BLK, 2)    // 2 blocks per CU (60 KB of LDS each): at most 256 VGPRs
void k_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                  uint32_t n, int shift, const uint32_t* __restrict__ hist,
                                                  uint32_t* status /* n_tiles rows of u16[RADIX], then (4 KB-aligned) one row of u32[RADIX] per group */, uint32_t n_tiles, uint32_t status_bytes,
                                                  int gshift, int use_ticket, uint32_t* tile_counter, uint32_t* err_flag) {
  __shared__ uint32_t s_keys[SORT_TILE];
  __shared__ uint32_t s_vals[SORT_TILE];
  __shared__ __align__(16) uint32_t s_whist[BLK / 64][RADIX];
  __shared__ __align__(16) uint32_t s_dstart[RADIX];
  __shared__ __align__(16) uint32_t s_gbase[RADIX];
  __shared__ __align__(16) uint32_t s_cnt[RADIX];
  __shared__ uint32_t s_wave[BLK / 64];
  __shared__ uint32_t s_tile;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Tile ids: handed out in start order when the grid is larger than what the chip holds at once (every predecessor of a
  // tile is then already running: the look-back cannot wait on a block that was never scheduled); the block index when
  // every tile is resident anyway (320 same-address ticket atomics spread the tile starts over 4 us).
  if (use_ticket && tid == 0) s_tile = atomicAdd(tile_counter, 1u);
  for (int i = tid; i < (BLK / 64) * RADIX / 4; i += BLK) reinterpret_cast<uint4*>(&s_whist[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
  reinterpret_cast<uint4*>(s_cnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  // (wave-uniform by construction; said so explicitly: everything derived from it - row addresses, the number of look-back
  // words to fetch - then lives in scalar registers, and rows that do not exist are skipped by a branch instead of predicated)
  const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(use_ticket ? s_tile : blockIdx.x));
  const uint32_t base = tile * (uint32_t)SORT_TILE;
  const uint32_t wbase = base + (uint32_t)wave * (64 * SORT_KPT);
  SORT_STAMP(0);

  uint32_t key[SORT_KPT], val[SORT_KPT], rank[SORT_KPT];
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
  }
  // the digit bases of this pass: exclusive scan of the global histogram (independent of everything below: issued early)
  const uint4 hcnt = reinterpret_cast<const uint4*>(hist)[tid];
  // 1. The tile's digit counts, as early as possible: every later tile waits for them (they are what the look-back sums),
  //    so they are counted with plain LDS atomics and published BEFORE the ranking, not derived from it.
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r)
    if ((wbase + r * 64 + lane) < n) atomicAdd(&s_cnt[(key[r] >> shift) & (uint32_t)(RADIX - 1)], 1u);
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {                         // payloads: needed at the scatter only
    uint32_t idx = wbase + r * 64 + lane;
    val[r] = FIRST ? idx : (idx < n ? vals_in[idx] : 0u);
  }
  __syncthreads();
  uint32_t total[DPT];
  {
    const uint4 c = reinterpret_cast<const uint4*>(s_cnt)[tid];    // thread t owns digits 4t .. 4t+3
    total[0] = c.x; total[1] = c.y; total[2] = c.z; total[3] = c.w;
  }
  // Look-back words, self-validating (a torn access is harmless: every word carries its own flag), moved with write-through
  // (sc1) buffer stores and sc1 loads:
  //   GROUP rows (4 KB each, behind the tile rows): one u32 per digit = 2-bit flag | 30-bit count; a thread's four digits
  //     travel as ONE 16-byte access;
  //   TILE rows (2 KB each): one u16 per digit = bit 15 "there" | the tile's count (<= SORT_TILE < 2^15); a thread's four
  //     digits are ONE 8-byte access, so all of a group's <= 31 predecessors fit the registers of a single round trip.
  static_assert(SORT_TILE < (1 << 15), "tile words are 16 bits");
  const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(status, 0, (int)status_bytes, 0x00020000);
  const uint32_t my_off = 16u * (uint32_t)tid;              // byte offset of my four words inside a group's 4 KB row
  const uint32_t group_rows = sort_group_rows_offset(n_tiles);
  auto publish = [&](uint32_t g, uint32_t f, const uint32_t v[DPT]) {
    const u32x4 x = {f | v[0], f | v[1], f | v[2], f | v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(x, st_rsrc, my_off, group_rows + g * (uint32_t)(RADIX * 4), 16);      // aux 16 = sc1
  };
  auto fetch = [&](uint32_t g) -> uint4 {
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(st_rsrc, my_off, group_rows + g * (uint32_t)(RADIX * 4), 16);   // (row: scalar offset)
    return make_uint4(x[0], x[1], x[2], x[3]);
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto fetch_tile = [&](uint32_t t) -> uint2 {
    const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(st_rsrc, 8u * (uint32_t)tid, t * (uint32_t)(RADIX * 2), 16);
    return make_uint2(x[0], x[1]);
  };
  {
    const u32x2 x = {0x80008000u | total[0] | (total[1] << 16), 0x80008000u | total[2] | (total[3] << 16)};
    __builtin_amdgcn_raw_buffer_store_b64(x, st_rsrc, 8u * (uint32_t)tid, tile * (uint32_t)(RADIX * 2), 16);
  }
  constexpr int LBG = 32;                                  // predecessors of my group fetched per round trip (a whole group of 32)
  constexpr int LB = 16;                                   // group words fetched per round trip
  const uint32_t grp = tile >> gshift, g0 = grp << gshift;
  const bool closes_group = ((tile + 1u) & ((1u << gshift) - 1u)) == 0u;
  uint32_t excl[DPT] = {0u, 0u, 0u, 0u};
  uint32_t spins = 0;
  bool fail = false;
  // (the nearest group words are fetched in the same round trip as the in-group words when `with_groups`: the walk below
  // starts from them)
  constexpr int PG = 8;
  uint4 pre_g[PG];
  bool have_pre_g = false;
  auto in_group_sum = [&](bool with_groups) {                              // counts of the predecessors p-1 .. g0 of my group: pure sums, any order
    for (uint32_t p = tile; p > g0 && !fail;) {
      const uint32_t nb = (p - g0 < (uint32_t)LBG) ? (p - g0) : (uint32_t)LBG;
      for (;;) {                                           // the whole batch again if any word is not there yet (never a stale register copy)
        uint2 w[LBG];
#pragma unroll
        for (int k = 0; k < LBG; ++k) if ((uint32_t)k < nb) w[k] = fetch_tile(p - 1u - k);
        if (with_groups && !have_pre_g && grp > 0u) {
#pragma unroll
          for (int k = 0; k < PG; ++k) if ((uint32_t)k < grp) pre_g[k] = fetch(grp - 1u - k);
          have_pre_g = true;
        }
        uint32_t all = ~0u;                                // all there <=> every half-word has its bit 15
#pragma unroll
        for (int k = 0; k < LBG; ++k) if ((uint32_t)k < nb) all &= w[k].x & w[k].y;
        if ((all & 0x80008000u) == 0x80008000u) {
          // four words are added as packed halves (4 x SORT_TILE < 2^16), then unpacked
#pragma unroll
          for (int c = 0; c < LBG; c += 4) {
            if ((uint32_t)c < nb) {
              uint32_t a = 0u, b = 0u;
#pragma unroll
              for (int k = c; k < c + 4; ++k) if ((uint32_t)k < nb) { a += w[k].x & 0x7FFF7FFFu; b += w[k].y & 0x7FFF7FFFu; }
              excl[0] += a & 0xFFFFu; excl[1] += a >> 16; excl[2] += b & 0xFFFFu; excl[3] += b >> 16;
            }
          }
          break;
        }
        if (++spins > SPIN_LIMIT) { atomicOr(err_flag, 1u); fail = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      p -= nb;
    }
  };
  // The tile that closes a group sums its group NOW (its predecessors publish their counts within the first microseconds
  // of their own run) and publishes the group's sum before it ranks: by the time the other tiles have ranked their keys,
  // every group word they need is there.  Group 0's sum is already its inclusive prefix.
  if (closes_group) {
    in_group_sum(false);
    const uint32_t v[DPT] = {excl[0] + total[0], excl[1] + total[1], excl[2] + total[2], excl[3] + total[3]};
    publish(grp, grp == 0u ? ST_INCL : ST_AGG, v);
  }
  SORT_STAMP(1);
  // 2. wave-level stable ranking: peers = lanes holding the same digit (10 ballots), rank inside
  // the peer group by popcount below the lane; the group leader bumps the wave's digit counter.
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    const bool valid = (wbase + r * 64 + lane) < n;
    const uint32_t d = (key[r] >> shift) & (uint32_t)(RADIX - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    // the wave's counter of digit d so far, read by every peer, then bumped by the group's first lane: LDS operations of one
    // wave execute in order and s_whist[wave] is this wave's own (no returning atomic + lane shuffle: one LDS round trip less)
    uint32_t prev = 0;
    if (valid) {
      prev = s_whist[wave][d];
      if (lane == __ffsll((long long)peers) - 1) atomicAdd(&s_whist[wave][d], (uint32_t)__popcll(peers));
    }
    rank[r] = prev + (uint32_t)__popcll(peers & lt_mask);
  }
  __syncthreads();
  {   // per-wave exclusive offsets of my four digits
    uint32_t run[DPT] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) {
      uint4* cell = reinterpret_cast<uint4*>(&s_whist[w][DPT * tid]);
      const uint4 c = *cell;
      *cell = make_uint4(run[0], run[1], run[2], run[3]);
      run[0] += c.x; run[1] += c.y; run[2] += c.z; run[3] += c.w;
    }
  }
  // Grouped look-back.  A chained walk (one predecessor per step, stop at the first inclusive prefix) takes ~25 dependent
  // hops when all 320 tiles of a 1.3 M-key pass run at once: 11 of a pass's 30 us.  Here tiles form groups of G = 2^gshift:
  //   1. every tile publishes its counts (AGG) before it ranks (above);
  //   2. the last tile of a group sums its group's counts and publishes the group's sum (AGG; group 0: INCL) in a second
  //      array, also before it ranks (above);
  //   3. after ranking (and after the tile-local scatter into LDS), every tile sums the counts of the predecessors INSIDE
  //      its group and walks back over the GROUP words (in order, until an INCL): the <= 31 tile words and the nearest 8 group
  //      words in one round trip, further group words 16 per round trip;
  //   4. the last tile of a group then publishes the group's inclusive prefix (INCL).
  // A tile only ever waits for tiles with smaller ids (already running).
  const uint32_t tsum = total[0] + total[1] + total[2] + total[3];
  const uint32_t dstart0 = block_excl_scan(tsum, s_wave, nullptr);
  const uint32_t dbase0 = block_excl_scan(hcnt.x + hcnt.y + hcnt.z + hcnt.w, s_wave, nullptr);
  SORT_STAMP(2);
  // LDS-staged scatter FIRST: a key's place in the tile-local sorted order needs the tile's own counts only, not the
  // look-back - and once the keys, payloads and ranks are in LDS their ~60 registers are free for the look-back words, so
  // the in-group words and the nearest group words travel in ONE round trip (they were two or three: the registers did not
  // allow more; with 64 + 32 registers of look-back words beside them the kernel spilled and a pass took 34 us instead of 25).
  {
    uint32_t ds = dstart0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) { s_dstart[DPT * tid + q] = ds; ds += total[q]; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    if ((wbase + r * 64 + lane) < n) {
      const uint32_t d = (key[r] >> shift) & (uint32_t)(RADIX - 1);
      const uint32_t pos = s_dstart[d] + s_whist[wave][d] + rank[r];
      s_keys[pos] = key[r];
      s_vals[pos] = val[r];
    }
  }
  SORT_STAMP(3);
  if (!closes_group) in_group_sum(true);                   // (their predecessors published before their own ranking: ready by now)
  if (__ballot(fail) != 0ull) fail = true;                 // (the walk below runs wave-wide or not at all)
  if (grp > 0u && !fail) {
    // group words in order, nearest first, until an inclusive prefix (group 0's is one); per digit word, since the four
    // words of a row may become visible one after the other
    // (The first version of this walk predicated all 4 x 16 (digit, word) steps per lane: ~1000 VALU instructions, 4.8 us of
    // a 25 us pass although a 256-tile pass has at most 7 group words.  Now the wave agrees on how far back anyone still has
    // to go, rows beyond that are skipped by a scalar branch, and a step is a handful of selects.)
    uint32_t pos[DPT] = {grp, grp, grp, grp};              // next group to consume is pos - 1; 0 = done
    for (;;) {
      const uint32_t top = max(max(pos[0], pos[1]), max(pos[2], pos[3]));
      const uint32_t utop = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(top));   // farthest any lane of the wave still has to go
      if (utop == 0u) break;
      uint4 w[LB];
      if (have_pre_g) {                                    // (only ever the first round: utop == grp) words 0 .. PG-1 came with the in-group words
#pragma unroll
        for (int k = 0; k < PG; ++k) w[k] = pre_g[k];
#pragma unroll
        for (int k = PG; k < LB; ++k) if ((uint32_t)k < utop) w[k] = fetch(utop - 1u - k);
        have_pre_g = false;
      } else {
#pragma unroll
        for (int k = 0; k < LB; ++k) if ((uint32_t)k < utop) w[k] = fetch(utop - 1u - k);   // word k = group utop-1-k
      }
      uint32_t taken[DPT] = {0u, 0u, 0u, 0u};
      bool act[DPT], done[DPT];
#pragma unroll
      for (int q = 0; q < DPT; ++q) { act[q] = pos[q] != 0u; done[q] = false; }
#pragma unroll
      for (int k = 0; k < LB; ++k) {
        if ((uint32_t)k < utop) {                          // scalar
#pragma unroll
          for (int q = 0; q < DPT; ++q) {
            const uint32_t x = q == 0 ? w[k].x : q == 1 ? w[k].y : q == 2 ? w[k].z : w[k].w, f = x >> 30;
            const bool on = act[q] && utop - (uint32_t)k == pos[q] - taken[q];      // word k is the next one digit q consumes
            const bool take = on && f != 0u;
            excl[q] += take ? (x & ST_MASK) : 0u;
            taken[q] += take ? 1u : 0u;
            done[q] = done[q] || (take && f == 2u);
            act[q] = act[q] && !(on && (f == 0u || f == 2u));
          }
        }
      }
      bool progressed = false;
#pragma unroll
      for (int q = 0; q < DPT; ++q) { progressed = progressed || taken[q] != 0u; pos[q] = done[q] ? 0u : pos[q] - taken[q]; }
      if (!progressed && top != 0u) {
        if (++spins > SPIN_LIMIT) { atomicOr(err_flag, 1u); pos[0] = pos[1] = pos[2] = pos[3] = 0u; }   // (no break: the loop is wave-uniform)
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (closes_group) {
      const uint32_t v[DPT] = {excl[0] + total[0], excl[1] + total[1], excl[2] + total[2], excl[3] + total[3]};
      publish(grp, ST_INCL, v);
    }
  }
  {
    const uint32_t hc[DPT] = {hcnt.x, hcnt.y, hcnt.z, hcnt.w};
    uint32_t ds = dstart0, db = dbase0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      s_gbase[DPT * tid + q] = db + excl[q] - ds;
      ds += total[q]; db += hc[q];
    }
  }
  __syncthreads();                                         // the tile's sorted keys and the digits' global bases are in LDS
  SORT_STAMP(4);
  // coalesced runs to HBM
  const uint32_t cnt = (n - base < (uint32_t)SORT_TILE) ? (n - base) : (uint32_t)SORT_TILE;
#pragma unroll
  for (int k = 0; k < SORT_KPT; ++k) {
    const uint32_t i = tid + k * BLK;
    if (i < cnt) {
      const uint32_t kk = s_keys[i];
      const uint32_t out = s_gbase[(kk >> shift) & (uint32_t)(RADIX - 1)] + i;
      keys_out[out] = kk;
      vals_out[out] = s_vals[i];
    }
  }
  SORT_STAMP(5);
}

// ---------------------------------------------------------------- Karras radix tree
// delta(i,j): common prefix of the augmented keys (code, index); -1 when j is out of range
RT_DEV int kdelta(const uint32_t* __restrict__ k, int n, int i, uint32_t ki, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t kj = k[j];
  if (ki != kj) return __clz((int)(ki ^ kj));
  return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// Node i of the tree: the other end j of its leaf range and its split position gamma (children [lo, gamma], [gamma + 1, hi]).
RT_DEV void karras_node(const uint32_t* __restrict__ k, int n, int i, int& j_out, int& gamma_out) {
  const uint32_t ki = k[i];
  const int d = (kdelta(k, n, i, ki, i + 1) - kdelta(k, n, i, ki, i - 1)) > 0 ? 1 : -1;
  const int dmin = kdelta(k, n, i, ki, i - d);
  int lmax = 2;
  while (kdelta(k, n, i, ki, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (kdelta(k, n, i, ki, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = kdelta(k, n, i, ki, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (kdelta(k, n, i, ki, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  j_out = j;
  gamma_out = i + s * d + min(d, 0);
}
// node ids: internal k -> k, leaf k -> (T-1)+k
RT_DEV void karras_children(int n, int i, int j, int gamma, int& c0, int& c1) {
  const int lo = min(i, j), hi = max(i, j);
  c0 = (lo == gamma) ? (n - 1) + gamma : gamma;
  c1 = (hi == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
}

// The same node from the same definition, for the nodes whose range crosses a 256-leaf block (k_refit_mid): their ranges
// can be long, and galloping + bisection is a chain of ~3 log2(range) dependent key loads scattered over the key array.
// delta(i, x) only shrinks as x moves away from i, so any probe sequence that brackets the last position with
// delta > threshold gives Karras' answer: the search is done by 16 lanes at once (16 probes per round trip: the root's
// ~80 dependent loads become ~14), and a probe 512 or more positions away is moved to the nearest multiple of 256, whose
// key comes from the compact table of every block's first key (20 KB at 1.3 M triangles) instead of a cold line of the key array.
struct KeyProbe {
  const uint32_t* __restrict__ k; const uint32_t* __restrict__ tb; int n; int i; uint32_t ki;
  RT_DEV int delta(int x) const {
    if (x < 0 || x >= n) return -1;
    const uint32_t kx = ((x & 255) == 0) ? tb[x >> 8] : k[x];
    return (ki != kx) ? __clz((int)(ki ^ kx)) : 32 + __clz((int)((uint32_t)i ^ (uint32_t)x));
  }
  // offset s moved to the multiple of 256 (as a position) nearest to i + s d that lies strictly between lo and hi, if any
  RT_DEV int snap(int d, int s, int lo, int hi) const {
    const int xs = ((i + s * d) + 128) & ~255;
    int ms = (xs - i) * d;
    if (ms <= lo) ms += 256; else if (ms >= hi) ms -= 256;
    return (ms > lo && ms < hi) ? ms : s;
  }
  // Largest s in [s_true, s_end) with delta(i + s d) > thr; delta(i + s_true d) > thr is given, s_end counts as failing.
  // By a GROUP of 16 lanes (gl = lane in the group; every lane holds the same node): sixteen probes per round trip - first
  // the doubling steps s_true + 2, 4, 8, ... (a short range ends in the first round, next to i), then 16-ary bisection.
  // delta is monotone, so a round's answers are a run of passes followed by a run of failures.
  RT_DEV int last_above(int lane, int d, int s_true, int s_end, int thr) const {
    const int gl = lane & 15, g0 = lane & 48;
    int lo = s_true, hi = s_end;
    for (int round = 0; round < 2; ++round) {           // gallop: steps 2^1 .. 2^16, then 2^17 .. 2^32 (clipped by s_end)
      const int sh = gl + 16 * round + 1;
      long long sl = (long long)s_true + (sh < 40 ? (1ll << sh) : (1ll << 40));
      const bool in = sl < (long long)hi;
      int sv = in ? (int)sl : hi;
      if (in && sh >= 9) sv = snap(d, sv, lo, hi);
      const bool ok = in && delta(i + sv * d) > thr;
      const uint32_t m = (uint32_t)__popc((uint32_t)((__ballot(ok) >> g0) & 0xFFFFull));     // passes are lanes 0 .. m-1
      if (m > 0u) lo = __shfl(sv, g0 + (int)m - 1);
      if (m < 16u) { hi = __shfl(sv, g0 + (int)m); break; }   // first failure (a lane beyond s_end holds hi itself)
    }
    while (hi - lo > 1) {                                // 15 interior probes per round
      const int len = hi - lo;
      int sv = lo + (int)(((long long)len * (gl + 1)) >> 4);
      if (len >= 16 * 512 && gl < 15) sv = snap(d, sv, lo, hi);
      const bool ok = gl < 15 && (sv == lo || delta(i + sv * d) > thr);
      const uint32_t m = (uint32_t)__popc((uint32_t)((__ballot(ok) >> g0) & 0x7FFFull));
      const int nlo = (m > 0u) ? __shfl(sv, g0 + (int)m - 1) : lo;
      const int nhi = (m < 15u) ? __shfl(sv, g0 + (int)m) : hi;
      lo = nlo; hi = nhi;
    }
    return lo;
  }
};
// by a group of 16 lanes that all hold the same i
RT_DEV void karras_node_wide(const uint32_t* __restrict__ k, const uint32_t* __restrict__ tb, int n, int i, int lane, int& j_out, int& gamma_out) {
  KeyProbe P; P.k = k; P.tb = tb; P.n = n; P.i = i; P.ki = k[i];
  const int d = (P.delta(i + 1) - P.delta(i - 1)) > 0 ? 1 : -1;
  const int dmin = P.delta(i - d);
  const int room = d > 0 ? (n - 1 - i) : i;             // positions on that side of i
  const int l = P.last_above(lane, d, 1, room + 1, dmin);   // delta(i, i + d) > dmin by the choice of d
  const int j = i + l * d;
  const int dnode = P.delta(j);
  const int sp = P.last_above(lane, d, 0, l, dnode);    // s = 0 is i itself
  j_out = j;
  gamma_out = i + sp * d + min(d, 0);
}

// ---------------------------------------------------------------- leaves + bottom-up refit
// No hand-off between workgroups inside a launch.  Node i of the Karras tree covers the contiguous leaf range
// [lo, hi] of the sorted order, and min / max are exact and order-independent, so a node's box is a RANGE query over the
// leaf boxes: nothing has to wait for its children.  Three launches, data crossing between them only at kernel boundaries:
//   k_refit_local   one thread per leaf: BVH-order triangle records and the leaf record; the Karras search of every node
//                   whose range lies inside the thread's 256-leaf block (keys + one halo key per side in LDS: 96.4 % of the
//                   nodes), their boxes by a climb through LDS, node + wide records written coalesced; the boxes of every
//                   16-leaf and 64-leaf group, the block's first key; nodes that cross the block boundary go to 64 shard lists
//   k_refit_mid     blocks [0, ns): per super-block of 256 blocks (65 536 leaves) a sparse table (9 levels) over its block
//                   boxes; the other blocks: the Karras search of the crossing nodes, 16 lanes per node (KeyProbe above)
//   k_refit_spans   the crossing nodes' boxes, 16 lanes per node: a child's box is read if the child is block-local or a
//                   leaf, else folded from <= 16 leaves + <= 3 groups of 16 + <= 3 groups of 64 at either ragged end of its
//                   range, <= 4 sparse-table entries and the boxes of whole super-blocks in between
// (Round 1: a separate k_karras, then a climb in which the ~1 % of nodes above the blocks were handed from CU to CU with
// write-through stores, an atomic counter and sc1 loads per level - a ~70 us latency chain over the top ~12 levels with the
// chip nearly idle, 2.7x the traffic, and a verify pass + fenced fallback to guard it.)
static_assert(BLK == 256, "block-local nodes: (lo >> 8) == (hi >> 8)");
constexpr int CROSS_SHARDS = 64;           // lists of block-crossing nodes (one counter per 128-byte line, block b appends to list b % 64)
constexpr int CROSS_STRIDE = 32;           // uint32 words between the shard counters
constexpr int SUPER = 256;                 // blocks per super-block
constexpr int ST_LEVELS = 9;               // 2^0 .. 2^8 blocks
constexpr float BOX_INF = __builtin_inff();

RT_DEV void store_node(float* nodes, int idx, V3 mn, float a, V3 mx, float b) {
  float4* nd = reinterpret_cast<float4*>(nodes) + 2 * (size_t)idx;
  nd[0] = make_float4(mn.x, mn.y, mn.z, a);
  nd[1] = make_float4(mx.x, mx.y, mx.z, b);
}
RT_DEV V3 min3(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
RT_DEV V3 max3(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
RT_DEV float wave_min_f(float v) { for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
RT_DEV float wave_max_f(float v) { for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
// 64-byte two-child record of node `par` (rt_internal.h), the statements of k_pack_wide
RT_DEV void store_wide(float4* wide, int par, int c0, int c1, int first_leaf, V3 a_mn, V3 a_mx, V3 b_mn, V3 b_mx) {
  const uint32_t r0 = (c0 >= first_leaf) ? (0x80000000u | (uint32_t)(c0 - first_leaf)) : (uint32_t)c0;
  const uint32_t r1 = (c1 >= first_leaf) ? (0x80000000u | (uint32_t)(c1 - first_leaf)) : (uint32_t)c1;
  wide[4 * (size_t)par + 0] = make_float4(a_mn.x, a_mn.y, a_mn.z, __uint_as_float(r0));
  wide[4 * (size_t)par + 1] = make_float4(a_mx.x, a_mx.y, a_mx.z, __uint_as_float(r1));
  wide[4 * (size_t)par + 2] = make_float4(b_mn.x, b_mn.y, b_mn.z, 0.0f);
  wide[4 * (size_t)par + 3] = make_float4(b_mx.x, b_mx.y, b_mx.z, 0.0f);
}

// N float4 per lane, lane l's at dst[l * N .. l * N + N): written as they stand, one store instruction of the wave covers 64
// pieces of 16 bytes spread over N KB, and the chip takes such stores at 4.4 TB/s; turned through the wave's LDS window so that
// every store instruction writes ONE contiguous 1 KB run it takes 5.6 TB/s (tools/ubench_stream, variants 0 / 4).
// `ok`: which lanes' records are to be written.  The window is the wave's own: LDS operations of one wave are in order, the
// fences only keep the compiler from moving them.
template <int N>
RT_DEV void wave_store_run(float4* win, float4* dst, const float4 (&v)[N], unsigned long long ok, int lane) {
#pragma unroll
  for (int i = 0; i < N; ++i) win[lane * N + i] = v[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int e = 64 * i + lane;
    if ((ok >> (e / N)) & 1ull) dst[e] = win[e];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                     // the window is rewritten by the next run
}

__global__ __launch_bounds__(BLK) void k_refit_local(const float4* __restrict__ mesh_rec,
                                                     const uint32_t* __restrict__ order, const uint32_t* __restrict__ keys, uint32_t n_tri,
                                                     int32_t* __restrict__ span,
                                                     float* __restrict__ nodes, float* __restrict__ tri_pos,
                                                     float* __restrict__ tri_nor, float4* __restrict__ wide, float4* __restrict__ tri_pre,
                                                     float4* __restrict__ tri_uv /* NULL: no surface records */,
                                                     float4* __restrict__ mini_box, float4* __restrict__ chunk_box,
                                                     uint32_t* __restrict__ block_key, uint32_t* cross_cnt, uint32_t* __restrict__ cross_list,
                                                     uint32_t cross_cap, bool v2) {
  // [0, BLK): this block's leaves, [BLK, 2 BLK): internal nodes with index in the block's range.  Before and after the climb
  // the same 12 KB are the waves' store windows (wave_store_run): 3 KB each
  __shared__ float4 s_win[2 * BLK * 6 / 4];
  float (*s_box)[6] = reinterpret_cast<float (*)[6]>(s_win);
  __shared__ uint32_t s_visit[BLK];        // arrival counters of the block-local internal nodes
  __shared__ int s_child[BLK][2];          // children of the block-local internal nodes
  __shared__ int s_par[2 * BLK];           // parent of every leaf / internal slot if that parent is block-local, else -1
  __shared__ uint32_t s_key[BLK + 2];      // sorted keys of positions S-1 .. S+256 (one halo key on either side)
  __shared__ uint32_t s_cross[BLK / 64 + 1];   // crossing nodes per wave (exclusive offsets after the scan), [4] = the block's base in its shard list
  const int n = (int)n_tri;
  const int S = (int)(blockIdx.x * BLK), E = min(S + BLK - 1, n - 1);   // this block's leaves [S, E]
  s_visit[threadIdx.x] = 0u;
  s_par[threadIdx.x] = -1; s_par[BLK + threadIdx.x] = -1;
  for (int q = threadIdx.x; q < BLK + 2; q += BLK) { const int p = S - 1 + q; if (p >= 0 && p < n) s_key[q] = keys[p]; }
  const uint32_t j = blockIdx.x * BLK + threadIdx.x;
  const bool valid = j < n_tri;
  const uint32_t ti = valid ? order[j] : 0u;
  __syncthreads();
  // the triangle's record (k_pack_mesh: its attributes, two 64-byte sectors) is requested here and used after the tree
  // search below, which lives in LDS: the search hides the fetch
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 r0 = zero4, r1 = zero4, r2 = zero4, r3 = zero4, r4 = zero4, r5 = zero4, r6 = zero4;
  if (valid) {
    const float4* rec = mesh_rec + (size_t)ti * MESH_REC;
    r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3]; r4 = rec[4];
    if (tri_uv) { r5 = rec[5]; r6 = rec[6]; }
  }
  if (threadIdx.x == 0) block_key[blockIdx.x] = s_key[1];   // key of position S: the coarse probes of k_refit_mid
  // Karras 2012 on the keys in LDS, for internal node i = j.  A probe outside [S, E] is only ever compared with dmin, and a
  // key farther away shares at most as long a prefix as the halo key of that side: if that halo key is still inside the
  // node's range (delta > dmin) the range crosses the block boundary and the node is left to k_refit_mid; if not, every
  // probe beyond it fails the comparison too.  Block-local nodes (~96 %) never leave LDS.
  const int first_leaf = n - 1;            // leaf j is node (T-1)+j, one triangle, start = j
  bool is_crossing = false;
  if ((int)j < n - 1) {
    const int i = (int)j;
    const uint32_t ki = s_key[i - (S - 1)];
    auto delta = [&](int x) -> int {                   // exact for x in [S-1, E+1]; -1 beyond (see above) and out of range
      if (x < S - 1 || x > E + 1 || x < 0 || x >= n) return -1;
      const uint32_t kx = s_key[x - (S - 1)];
      return (ki != kx) ? __clz((int)(ki ^ kx)) : 32 + __clz((int)((uint32_t)i ^ (uint32_t)x));
    };
    const int d = (delta(i + 1) - delta(i - 1)) > 0 ? 1 : -1;
    const int dmin = delta(i - d);
    int other = -1;
    if (!(delta(d > 0 ? E + 1 : S - 1) > dmin)) {      // the range ends inside the block
      int lmax = 2;
      while (delta(i + lmax * d) > dmin) lmax <<= 1;
      int l = 0;
      for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(i + (l + t) * d) > dmin) l += t;
      other = i + l * d;
      const int dnode = delta(other);
      int sp = 0;
      for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(i + (sp + t) * d) > dnode) sp += t;
        if (t == 1) break;
      }
      const int gamma = i + sp * d + min(d, 0);
      int c0, c1;
      karras_children(n, i, other, gamma, c0, c1);
      s_child[threadIdx.x][0] = c0; s_child[threadIdx.x][1] = c1;
      s_par[(c0 >= first_leaf) ? (c0 - first_leaf - S) : (BLK + c0 - S)] = i;
      s_par[(c1 >= first_leaf) ? (c1 - first_leaf - S) : (BLK + c1 - S)] = i;
    }
    span[i] = other;                                   // -1: crosses a block boundary
    is_crossing = other < 0;
  }
  // the crossing nodes of the chip as 64 compact lists (k_refit_search / k_refit_spans then spend their lanes on those 3-4 %
  // of the nodes only): one atomic per block reserves the block's run in list (block % 64)
  const unsigned long long cross_mask = __ballot(is_crossing);
  if ((threadIdx.x & 63) == 0) s_cross[threadIdx.x >> 6] = (uint32_t)__popcll(cross_mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < BLK / 64; ++w) { const uint32_t c = s_cross[w]; s_cross[w] = run; run += c; }
    s_cross[BLK / 64] = run ? atomicAdd(cross_cnt + (blockIdx.x % CROSS_SHARDS) * CROSS_STRIDE, run) : 0u;
  }
  V3 mn = v3(BOX_INF, BOX_INF, BOX_INF), mx = v3(-BOX_INF, -BOX_INF, -BOX_INF);
  {
    const int lane = threadIdx.x & 63;
    float4* win = s_win + (threadIdx.x >> 6) * (3 * 64);               // this wave's 3 KB
    const size_t wj = j - (uint32_t)lane;                             // the wave's first triangle: its runs in the sorted arrays start there
    const unsigned long long ok = __ballot(valid);
    const V3 p[3] = {v3(r0.x, r0.y, r0.z), v3(r0.w, r1.x, r1.y), v3(r1.z, r1.w, r2.x)};
    const V3 nn[3] = {v3(r2.y, r2.z, r2.w), v3(r3.x, r3.y, r3.z), v3(r3.w, r4.x, r4.y)};
    {                                                  // web/script.js:530-545
      const float4 tp[3] = {make_float4(p[0].x, p[0].y, p[0].z, 0.0f), make_float4(p[1].x, p[1].y, p[1].z, 0.0f), make_float4(p[2].x, p[2].y, p[2].z, 0.0f)};
      wave_store_run<3>(win, reinterpret_cast<float4*>(tri_pos) + 3 * wj, tp, ok, lane);
      const float4 tn[3] = {make_float4(nn[0].x, nn[0].y, nn[0].z, 0.0f), make_float4(nn[1].x, nn[1].y, nn[1].z, 0.0f), make_float4(nn[2].x, nn[2].y, nn[2].z, 0.0f)};
      wave_store_run<3>(win, reinterpret_cast<float4*>(tri_nor) + 3 * wj, tn, ok, lane);
    }
    {                                                  // traversal layout of the triangle, the statements of k_pack_tris
      const V3 e1 = p[1] - p[0], e2 = p[2] - p[0];
      const V3 ng = cross3(e1, e2);
      const float4 pre[3] = {make_float4(p[0].x, p[0].y, p[0].z, ng.x), make_float4(e1.x, e1.y, e1.z, ng.y), make_float4(e2.x, e2.y, e2.z, ng.z)};
      wave_store_run<3>(win, tri_pre + 3 * wj, pre, ok, lane);
    }
    if (tri_uv) {                                      // surface record (rtcore.h "material / texture shading"), as k_gather_uv
      const float4 uv[2] = {make_float4(r4.z, r4.w, r5.x, r5.y), make_float4(r5.z, r5.w, r6.x, 0.0f)};
      wave_store_run<2>(win, tri_uv + 2 * wj, uv, ok, lane);
    }
    if (valid) {                                       // true triangle bounds (web/script.js:309-316)
      mn = v3(fminf(fminf(p[0].x, p[1].x), p[2].x), fminf(fminf(p[0].y, p[1].y), p[2].y), fminf(fminf(p[0].z, p[1].z), p[2].z));
      mx = v3(fmaxf(fmaxf(p[0].x, p[1].x), p[2].x), fmaxf(fmaxf(p[0].y, p[1].y), p[2].y), fmaxf(fmaxf(p[0].z, p[1].z), p[2].z));
    }
    // leaf record: A = tri_start, B = -count (script.js:348,554,558), the words of store_node
    const float4 leaf[2] = {make_float4(mn.x, mn.y, mn.z, enc_index(v2, j)), make_float4(mx.x, mx.y, mx.z, enc_leaf_b(v2, 1u))};
    wave_store_run<2>(win, reinterpret_cast<float4*>(nodes) + 2 * ((size_t)first_leaf + wj), leaf, ok, lane);
  }
  __syncthreads();                                     // the windows become the climb's boxes
  {   // boxes of aligned groups of 16 and 64 leaves (the valid ones), from one xor butterfly: the ragged ends of the ranges in k_refit_spans
    const int lane = threadIdx.x & 63;
    V3 a = mn, b = mx;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { a = min3(a, v3(__shfl_xor(a.x, o), __shfl_xor(a.y, o), __shfl_xor(a.z, o))); b = max3(b, v3(__shfl_xor(b.x, o), __shfl_xor(b.y, o), __shfl_xor(b.z, o))); }
    if ((lane & 15) == 0) {
      float4* q = mini_box + 2 * ((size_t)blockIdx.x * (BLK / 16) + (threadIdx.x >> 4));
      q[0] = make_float4(a.x, a.y, a.z, 0.0f); q[1] = make_float4(b.x, b.y, b.z, 0.0f);
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) { a = min3(a, v3(__shfl_xor(a.x, o), __shfl_xor(a.y, o), __shfl_xor(a.z, o))); b = max3(b, v3(__shfl_xor(b.x, o), __shfl_xor(b.y, o), __shfl_xor(b.z, o))); }
    if (lane == 0) {
      float4* q = chunk_box + 2 * ((size_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6));
      q[0] = make_float4(a.x, a.y, a.z, 0.0f); q[1] = make_float4(b.x, b.y, b.z, 0.0f);
    }
  }
  // The climb itself touches LDS only: boxes and arrival counters.  The records of the nodes it finishes are written
  // afterwards, thread t = internal node (block start + t): consecutive threads, consecutive 32- and 64-byte records.
  int cur = first_leaf + (int)j;
  int cur_slot = (int)threadIdx.x;                     // LDS slot of `cur`
  {
    float* sb = s_box[cur_slot];
    sb[0] = mn.x; sb[1] = mn.y; sb[2] = mn.z; sb[3] = mx.x; sb[4] = mx.y; sb[5] = mx.z;
  }
  int pe = s_par[threadIdx.x];
  while (pe >= 0) {                                    // -1: the parent's range leaves the block (k_refit_mid), or this is the root
    const int par = pe;
    __threadfence_block();                             // the box is in LDS before the counter moves (one LDS, in-order)
    const uint32_t arrived = atomicAdd(&s_visit[par & (BLK - 1)], 1u);
    if (arrived == 0) break;                           // the sibling's thread will finish this node
    const int c0 = s_child[par & (BLK - 1)][0], c1 = s_child[par & (BLK - 1)][1];
    const int sib = (c0 == cur) ? c1 : c0;
    __threadfence_block();
    const int sslot = (sib >= first_leaf) ? ((sib - first_leaf) & (BLK - 1)) : (BLK + (sib & (BLK - 1)));
    const float* ss = s_box[sslot];
    mn = min3(mn, v3(ss[0], ss[1], ss[2])); mx = max3(mx, v3(ss[3], ss[4], ss[5]));
    pe = s_par[BLK + (par & (BLK - 1))];
    cur = par;
    cur_slot = BLK + (par & (BLK - 1));
    float* sb = s_box[cur_slot];
    sb[0] = mn.x; sb[1] = mn.y; sb[2] = mn.z; sb[3] = mx.x; sb[4] = mx.y; sb[5] = mx.z;
  }
  __syncthreads();
  {
    const bool done = j + 1 < n_tri && s_visit[threadIdx.x] == 2u;   // internal node j was finished here: both children are in LDS
    float4 w[4] = {zero4, zero4, zero4, zero4}, nd[2] = {zero4, zero4};
    if (done) {
      const int c0 = s_child[threadIdx.x][0], c1 = s_child[threadIdx.x][1];
      const float* a = s_box[(c0 >= first_leaf) ? ((c0 - first_leaf) & (BLK - 1)) : (BLK + (c0 & (BLK - 1)))];
      const float* b = s_box[(c1 >= first_leaf) ? ((c1 - first_leaf) & (BLK - 1)) : (BLK + (c1 & (BLK - 1)))];
      const float* m = s_box[BLK + threadIdx.x];
      // the words of store_wide and store_node
      const uint32_t q0 = (c0 >= first_leaf) ? (0x80000000u | (uint32_t)(c0 - first_leaf)) : (uint32_t)c0;
      const uint32_t q1 = (c1 >= first_leaf) ? (0x80000000u | (uint32_t)(c1 - first_leaf)) : (uint32_t)c1;
      w[0] = make_float4(a[0], a[1], a[2], __uint_as_float(q0)); w[1] = make_float4(a[3], a[4], a[5], __uint_as_float(q1));
      w[2] = make_float4(b[0], b[1], b[2], 0.0f); w[3] = make_float4(b[3], b[4], b[5], 0.0f);
      nd[0] = make_float4(m[0], m[1], m[2], enc_index(v2, (uint32_t)c0)); nd[1] = make_float4(m[3], m[4], m[5], enc_index(v2, (uint32_t)c1));   // v1: indices as f32 values (F11)
    }
    __syncthreads();                                   // every box has been read: the 12 KB are store windows again
    const int lane = threadIdx.x & 63;
    float4* win = s_win + (threadIdx.x >> 6) * (3 * 64);
    const size_t wj = j - (uint32_t)lane;
    const unsigned long long ok = __ballot(done);      // the other ~4 % are written by k_refit_spans
    wave_store_run<2>(win, reinterpret_cast<float4*>(nodes) + 2 * wj, nd, ok, lane);
#pragma unroll
    for (int half = 0; half < 2; ++half) {             // 64-byte records: 32 lanes' worth (2 KB) fit the window
      if ((lane >> 5) == half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) win[(lane & 31) * 4 + i] = w[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = 64 * i + lane;
        if ((ok >> (32 * half + e / 4)) & 1ull) wide[4 * (wj + 32 * half) + e] = win[e];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (is_crossing) {
    const int lane = threadIdx.x & 63;
    const uint32_t at = s_cross[BLK / 64] + s_cross[threadIdx.x >> 6] + (uint32_t)__popcll(cross_mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    cross_list[(size_t)(blockIdx.x % CROSS_SHARDS) * cross_cap + at] = j;
  }
}

// Two independent jobs in one launch (neither is large enough to fill the chip):
//  blocks [0, n_super): sparse table over the boxes of one super-block's 256 blocks: st[(s * 9 + k) * 256 + x] = box of blocks
//    [x, x + 2^k) of super-block s (entries reaching past block 255 are never queried); missing blocks count as empty boxes;
//  the other blocks: the tree search of the block-crossing nodes from their compact lists, 16 lanes per node (they are 3-4 %
//    of the nodes: searched inside a kernel with one lane per node, every wave waited for its two or three long searches
//    with 60 idle lanes, five rounds of blocks deep): wave -> shard list (wave % 64), four entries per step.
__global__ __launch_bounds__(SUPER) void k_refit_mid(const float4* __restrict__ chunk_box, uint32_t n_chunks, uint32_t n_blocks, uint32_t n_super,
                                                     float4* __restrict__ st, float4* __restrict__ super_box,
                                                     const uint32_t* __restrict__ keys, const uint32_t* __restrict__ block_key, uint32_t n_tri,
                                                     const uint32_t* __restrict__ cross_cnt, const uint32_t* __restrict__ cross_list, uint32_t cross_cap,
                                                     int32_t* __restrict__ span, int32_t* __restrict__ split) {
  __shared__ float s_lvl[2][SUPER][6];
  const int t = threadIdx.x;
  if (blockIdx.x >= n_super) {
    const uint32_t gw = (blockIdx.x - n_super) * (SUPER / 64) + (uint32_t)(t >> 6), n_waves = (gridDim.x - n_super) * (SUPER / 64);
    const uint32_t shard = gw % CROSS_SHARDS, cnt = cross_cnt[shard * CROSS_STRIDE];
    for (uint32_t q = (gw / CROSS_SHARDS) * 4u + (uint32_t)((t & 63) >> 4); q < cnt; q += (n_waves / CROSS_SHARDS) * 4u) {   // 16 lanes per node
      const int i = (int)cross_list[(size_t)shard * cross_cap + q];
      int j, gamma;
      karras_node_wide(keys, block_key, (int)n_tri, i, t & 63, j, gamma);
      if ((t & 15) == 0) { span[i] = j; split[i] = gamma; }
    }
    return;
  }
  const uint32_t s = blockIdx.x, b = s * SUPER + (uint32_t)t;
  V3 mn = v3(BOX_INF, BOX_INF, BOX_INF), mx = v3(-BOX_INF, -BOX_INF, -BOX_INF);
  if (b < n_blocks)
    for (uint32_t q = 4u * b; q < 4u * b + 4u && q < n_chunks; ++q) { mn = min3(mn, xyz(chunk_box[2 * (size_t)q])); mx = max3(mx, xyz(chunk_box[2 * (size_t)q + 1])); }
  int cur = 0;
  for (int k = 0; k < ST_LEVELS; ++k) {
    float4* out = st + (((size_t)s * ST_LEVELS + k) * SUPER + t) * 2;
    out[0] = make_float4(mn.x, mn.y, mn.z, 0.0f); out[1] = make_float4(mx.x, mx.y, mx.z, 0.0f);
    if (k == ST_LEVELS - 1) break;
    float* me = s_lvl[cur][t];
    me[0] = mn.x; me[1] = mn.y; me[2] = mn.z; me[3] = mx.x; me[4] = mx.y; me[5] = mx.z;
    __syncthreads();
    const int h = 1 << k;
    if (t + h < SUPER) {
      const float* o = s_lvl[cur][t + h];
      mn = min3(mn, v3(o[0], o[1], o[2])); mx = max3(mx, v3(o[3], o[4], o[5]));
    }
    cur ^= 1;                                          // the other buffer is written next level: no second barrier needed
  }
  if (t == 0) { super_box[2 * (size_t)s] = make_float4(mn.x, mn.y, mn.z, 0.0f); super_box[2 * (size_t)s + 1] = make_float4(mx.x, mx.y, mx.z, 0.0f); }
}

struct SpanCtx { const float4* nodes; const float4* mini_box; const float4* chunk_box; const float4* st; const float4* super_box; int first_leaf; };
RT_DEV void fold(V3& mn, V3& mx, const float4 a, const float4 b) { mn = min3(mn, xyz(a)); mx = max3(mx, xyz(b)); }

// What a 16-lane group fetches for ONE child [a, b] of a crossing node, five independent record loads per lane.
// A child that does not cross a block boundary itself (a leaf, or a node k_refit_local finished) is one record: nodes[id].
// Otherwise the range is folded from
//   head: the <= 16 leaves of a's 16-leaf group from a on, the <= 3 whole 16-groups up to the end of a's 64-leaf chunk,
//         the <= 3 whole chunks up to the end of a's block
//   tail: the mirror image in front of b
//   the whole blocks in between: <= 4 sparse-table entries (two overlapping ones per super-block touched) and, for the
//         handful of ranges over more than two super-blocks, the whole super-blocks' boxes (16 per round)
struct SpanLoads { float4 r[5][2]; uint32_t s1, s2; };
RT_DEV void span_issue(const SpanCtx& c, uint32_t gl, bool local, int id, uint32_t a, uint32_t b, SpanLoads& L) {
  const float4* leaf = c.nodes + 2 * (size_t)c.first_leaf;
  const float4 e0 = make_float4(BOX_INF, BOX_INF, BOX_INF, 0.0f), e1 = make_float4(-BOX_INF, -BOX_INF, -BOX_INF, 0.0f);
  bool ok[5];
  const float4* p[5];
  {   // head
    const uint32_t k = (a & ~15u) + gl;
    ok[0] = local ? (gl == 0u) : (k >= a);
    p[0] = local ? (c.nodes + 2 * (size_t)id) : (leaf + 2 * (size_t)(k >= a ? k : a));
    const bool mini = gl < 3u;
    const uint32_t q = mini ? ((a >> 4) + 1u + gl) : ((a >> 6) + 1u + (gl - 3u));
    ok[1] = !local && gl < 6u && (mini ? ((q >> 2) == (a >> 6)) : ((q >> 2) == (a >> 8)));
    p[1] = (mini ? c.mini_box : c.chunk_box) + 2 * (size_t)(ok[1] ? q : 0u);
  }
  {   // tail
    const uint32_t k = (b & ~15u) + gl;
    ok[2] = !local && k <= b;
    p[2] = leaf + 2 * (size_t)(ok[2] ? k : b);
    const bool mini = gl < 3u;
    const uint32_t q = mini ? (((b >> 6) << 2) + gl) : (((b >> 8) << 2) + (gl - 3u));
    ok[3] = !local && gl < 6u && (mini ? (q < (b >> 4)) : (q < (b >> 6)));
    p[3] = (mini ? c.mini_box : c.chunk_box) + 2 * (size_t)(ok[3] ? q : 0u);
  }
  {   // whole blocks m1 .. m2 in between: lanes 0-1 / 2-3 take one super-block's two entries each
    const uint32_t b1 = a >> 8, b2 = b >> 8;
    const bool any = !local && b2 > b1 + 1u;
    const uint32_t m1 = b1 + 1u, m2 = any ? b2 - 1u : m1;
    L.s1 = m1 >> 8; L.s2 = any ? (m2 >> 8) : L.s1;
    const bool second = gl >= 2u;                       // lanes 2, 3: the last super-block touched (if it is another one)
    const uint32_t sb = second ? L.s2 : L.s1;
    const uint32_t xx = second ? 0u : (m1 & 255u);
    const uint32_t yy = (second || L.s1 == L.s2) ? (m2 & 255u) : 255u;
    ok[4] = any && gl < 4u && (!second || L.s1 != L.s2);
    const uint32_t kk = 31u - (uint32_t)__clz((int)((ok[4] ? yy - xx : 0u) + 1u));
    const uint32_t e = (gl & 1u) ? (yy + 1u - (1u << kk)) : xx;
    p[4] = c.st + (ok[4] ? (((size_t)sb * ST_LEVELS + kk) * SUPER + e) * 2 : 0);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) { L.r[k][0] = p[k][0]; L.r[k][1] = p[k][1]; }
#pragma unroll
  for (int k = 0; k < 5; ++k) if (!ok[k]) { L.r[k][0] = e0; L.r[k][1] = e1; }
}
RT_DEV void span_fold(const SpanCtx& c, uint32_t gl, const SpanLoads& L, V3& omn, V3& omx) {
  V3 mn = xyz(L.r[0][0]), mx = xyz(L.r[0][1]);
#pragma unroll
  for (int k = 1; k < 5; ++k) fold(mn, mx, L.r[k][0], L.r[k][1]);
  for (uint32_t s = L.s1 + 1u + gl; s < L.s2; s += 16u) fold(mn, mx, c.super_box[2 * (size_t)s], c.super_box[2 * (size_t)s + 1]);   // whole super-blocks
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {                   // the 16-lane group's reduction; the result is in each of its lanes
    mn = min3(mn, v3(__shfl_xor(mn.x, o), __shfl_xor(mn.y, o), __shfl_xor(mn.z, o)));
    mx = max3(mx, v3(__shfl_xor(mx.x, o), __shfl_xor(mx.y, o), __shfl_xor(mx.z, o)));
  }
  omn = mn; omx = mx;
}

// The boxes and records of the block-crossing nodes, from the shard lists: 16 lanes per node (four nodes per wave and
// step), both children fetched in one round trip.  span / split: the node's range end and split position (k_refit_mid).
__global__ __launch_bounds__(BLK) void k_refit_spans(const uint32_t* __restrict__ cross_cnt, const uint32_t* __restrict__ cross_list, uint32_t cross_cap,
                                                     const int32_t* __restrict__ span, const int32_t* __restrict__ split, uint32_t n_tri,
                                                     float* nodes, float4* __restrict__ wide, const float4* __restrict__ mini_box,
                                                     const float4* __restrict__ chunk_box, const float4* __restrict__ st,
                                                     const float4* __restrict__ super_box, bool v2) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6), n_waves = gridDim.x * (BLK / 64);
  const uint32_t shard = gw % CROSS_SHARDS, cnt = cross_cnt[shard * CROSS_STRIDE];
  SpanCtx c; c.nodes = reinterpret_cast<const float4*>(nodes); c.mini_box = mini_box; c.chunk_box = chunk_box; c.st = st; c.super_box = super_box;
  c.first_leaf = (int)n_tri - 1;
  const uint32_t gl = (uint32_t)lane & 15u;
  for (uint32_t q = (gw / CROSS_SHARDS) * 4u + (uint32_t)(lane >> 4); q < cnt; q += (n_waves / CROSS_SHARDS) * 4u) {
    const int ii = (int)cross_list[(size_t)shard * cross_cap + q];
    const int jj = span[ii], gm = split[ii];
    int k0, k1;
    karras_children((int)n_tri, ii, jj, gm, k0, k1);
    const uint32_t lo = (uint32_t)min(ii, jj), hi = (uint32_t)max(ii, jj), gamma = (uint32_t)gm;   // children: [lo, gamma] and [gamma + 1, hi]
    SpanLoads A, B;
    span_issue(c, gl, (lo >> 8) == (gamma >> 8), k0, lo, gamma, A);
    span_issue(c, gl, ((gamma + 1u) >> 8) == (hi >> 8), k1, gamma + 1u, hi, B);
    V3 amn, amx, bmn, bmx;
    span_fold(c, gl, A, amn, amx);
    span_fold(c, gl, B, bmn, bmx);
    if (gl == 0u) {
      store_wide(wide, ii, k0, k1, c.first_leaf, amn, amx, bmn, bmx);
      store_node(nodes, ii, min3(amn, bmn), enc_index(v2, (uint32_t)k0), max3(amx, bmx), enc_index(v2, (uint32_t)k1));
    }
  }
}

// ---------------------------------------------------------------- traversal layout
RT_DEV uint32_t make_ref(const float4 a, const float4 b, int node, int leaf_shift, bool v2, uint32_t n_tri, uint32_t* err) {
  if (dec_is_leaf(v2, b.w)) {                         // leaf test of fetchCachedNode, compute.wgsl:75
    const uint32_t start = (uint32_t)dec_index(v2, a.w), count = (uint32_t)dec_count(v2, b.w);
    const uint32_t max_start = (leaf_shift >= 31) ? 0x7FFFFFFFu : ((1u << leaf_shift) - 1u);
    const uint32_t max_count = (leaf_shift >= 31) ? 1u : (1u << (31 - leaf_shift));
    if (start > max_start || count < 1 || count > max_count) { atomicOr(err, 2u); return 0xFFFFFFFFu; }
    // the leaf's triangles must exist: k_render reads tri_pre / tri_nor [start, start + count) (adopted device buffers
    // never pass the host's validate_nodes)
    if ((unsigned long long)start + count > n_tri) { atomicOr(err, 4u); return 0xFFFFFFFFu; }
    return 0x80000000u | ((count - 1u) << leaf_shift) | start;
  }
  return (uint32_t)node;
}

__global__ __launch_bounds__(BLK) void k_pack_wide(const float* __restrict__ nodes, uint32_t n_nodes, float4* __restrict__ wide,
                                                   uint32_t n_wide, int leaf_shift, bool v2, uint32_t n_tri, uint32_t* root_ref,
                                                   uint32_t* err) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_nodes) return;
  const float4* nd = reinterpret_cast<const float4*>(nodes);
  const float4 a = nd[2 * (size_t)i], b = nd[2 * (size_t)i + 1];
  if (i == 0) *root_ref = make_ref(a, b, 0, leaf_shift, v2, n_tri, err);
  if (dec_is_leaf(v2, b.w)) return;
  const int c0 = dec_index(v2, a.w), c1 = dec_index(v2, b.w);
  if (i >= n_wide || c0 < 0 || c1 < 0 || (uint32_t)c0 >= n_nodes || (uint32_t)c1 >= n_nodes) { atomicOr(err, 4u); return; }
  const float4 a0 = nd[2 * (size_t)c0], b0 = nd[2 * (size_t)c0 + 1];
  const float4 a1 = nd[2 * (size_t)c1], b1 = nd[2 * (size_t)c1 + 1];
  const uint32_t r0 = make_ref(a0, b0, c0, leaf_shift, v2, n_tri, err), r1 = make_ref(a1, b1, c1, leaf_shift, v2, n_tri, err);
  wide[4 * (size_t)i + 0] = make_float4(a0.x, a0.y, a0.z, __uint_as_float(r0));
  wide[4 * (size_t)i + 1] = make_float4(b0.x, b0.y, b0.z, __uint_as_float(r1));
  wide[4 * (size_t)i + 2] = make_float4(a1.x, a1.y, a1.z, 0.0f);
  wide[4 * (size_t)i + 3] = make_float4(b1.x, b1.y, b1.z, 0.0f);
}

__global__ __launch_bounds__(BLK) void k_pack_tris(const float* __restrict__ tri_pos, uint32_t n_tri, float4* __restrict__ tri_pre) {
  const uint32_t t = blockIdx.x * BLK + threadIdx.x;
  if (t >= n_tri) return;
  const float4* tp = reinterpret_cast<const float4*>(tri_pos) + (size_t)t * 3;
  const V3 v0 = xyz(tp[0]), v1 = xyz(tp[1]), v2 = xyz(tp[2]);
  const V3 e1 = v1 - v0, e2 = v2 - v0;               // compute.wgsl:114-115 == :282 operands
  const V3 ng = cross3(e1, e2);
  tri_pre[3 * (size_t)t + 0] = make_float4(v0.x, v0.y, v0.z, ng.x);
  tri_pre[3 * (size_t)t + 1] = make_float4(e1.x, e1.y, e1.z, ng.y);
  tri_pre[3 * (size_t)t + 2] = make_float4(e2.x, e2.y, e2.z, ng.z);
}

}  // namespace

#ifdef RT_SORT_TIMELINE
extern "C" int rt_debug_sort_timeline(unsigned long long* out, size_t n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_tl), sizeof(unsigned long long) * n_words);
}
#endif

uint32_t rt_sort_tile_count(uint32_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
// Tiles the chip certainly holds at once (occupancy query x CUs, with one block per CU of margin: the API can answer one
// block per CU high, MI355X_MICROARCH.md "Residency"); grids up to this size need no start-order tickets.
static uint32_t sort_resident_tiles() {
  static int cached = -1;
  if (cached < 0) {
    int dev = 0, n_cu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_onesweep<false>, BLK, 0) != hipSuccess || nb < 2) cached = 0;
    else cached = (nb - 1) * n_cu;
  }
  return (uint32_t)cached;
}
// small sorts: all tiles run at once and latency matters (groups of 32: one or two round trips inside the group, one over the
// groups); larger ones: groups of 16.  Sort time at 1.31 M / 10.5 M / 99.6 M keys with groups of 8: 90 us / 0.43 ms / 3.29 ms,
// of 16: 79 / 0.345 / 3.14, of 32: 77 / 0.365 / -, of 64: 81 / - / -
static int sort_group_shift(uint32_t tiles) { return tiles <= 512u ? 5 : 4; }
static uint32_t sort_group_count(uint32_t tiles) { const int sh = sort_group_shift(tiles); return (tiles + (1u << sh) - 1u) >> sh; }
// u32 words (used as 8-byte pairs): per pass the tiles' words, then the groups' words
static size_t sort_pass_status_bytes(uint32_t tiles) { return (size_t)sort_group_rows_offset(tiles) + (size_t)sort_group_count(tiles) * RADIX * 4; }
size_t rt_sort_status_words(uint32_t n) { return (size_t)SORT_PASSES * sort_pass_status_bytes(rt_sort_tile_count(n)) / 4; }
uint32_t rt_sort_hist_words() { return (uint32_t)(SORT_PASSES * RADIX); }

uint32_t rt_bounds_words() { return 6u * BOUNDS_MAX_BLOCKS; }   // the block cap x 6 partials
void rt_launch_bounds(const RtBuildBuffers& b, hipStream_t s) {
  const int g = grid_for(b.n_tri, BLK, BOUNDS_MAX_BLOCKS);           // = bounds_grid(n_tri): k_morton folds that many partials
  hipLaunchKernelGGL(k_bounds, dim3(g), dim3(BLK), 0, s, b.tri_in_pos, b.n_tri, b.bounds_enc, b.centroids, b.hist, b.tile_counter);
}
static int bounds_grid(uint32_t n_tri) { return grid_for(n_tri, BLK, BOUNDS_MAX_BLOCKS); }

size_t rt_mesh_rec_float4s(uint32_t n_tri) { return (size_t)n_tri * MESH_REC; }
void rt_launch_pack_mesh(const float* pos, const float* nor, const uint32_t* vidx, const uint32_t* nidx, const float* texcoords,
                         const uint32_t* tex_idx, const int32_t* material_ids, uint32_t n_tri, float4* mesh_rec, float* tri_in_pos, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_mesh, dim3((n_tri + BLK - 1) / BLK), dim3(BLK), 0, s, pos, nor, vidx, nidx, texcoords, tex_idx, material_ids, n_tri, mesh_rec, tri_in_pos);
}

void rt_launch_morton(const RtBuildBuffers& b, hipStream_t s) {
  // few, fat blocks: every block flushes its LDS digit histograms with one global atomic per non-empty bin
  hipLaunchKernelGGL(k_morton, dim3(grid_for(b.n_tri, BLK * 16, 512)), dim3(BLK), 0, s, b.centroids, b.n_tri, b.bounds_enc, (uint32_t)bounds_grid(b.n_tri), b.bounds_f, b.codes,
                     b.hist, b.tile_status, rt_sort_status_words(b.n_tri));
}

void rt_launch_sort(const RtBuildBuffers& b, hipStream_t s) {
  const uint32_t n = b.n_tri, tiles = rt_sort_tile_count(n);
  // hist / tile_counter were zeroed by k_bounds, tile_status by k_morton, which also filled hist
  // codes -> a -> b -> a: the sorted keys / order end up in keys_a / vals_a
  const uint32_t* ki = b.codes; const uint32_t* vi = nullptr;
  uint32_t *ko = b.keys_a, *vo = b.vals_a;
  for (int p = 0; p < SORT_PASSES; ++p) {
    const uint32_t st_bytes = (uint32_t)sort_pass_status_bytes(tiles);       // < 2^31: 52 429 tiles x 2 KB + their groups at 2^28 keys
    uint32_t* st = b.tile_status + (size_t)p * (st_bytes / 4);
    const int gsh = sort_group_shift(tiles);
    const int ticket = tiles > sort_resident_tiles() ? 1 : 0;
    if (p == 0)
      hipLaunchKernelGGL(k_onesweep<true>, dim3(tiles), dim3(BLK), 0, s, ki, vi, ko, vo, n, RADIX_BITS * p, b.hist + p * RADIX, st, tiles, st_bytes, gsh, ticket, b.tile_counter + p, b.err_flag);
    else
      hipLaunchKernelGGL(k_onesweep<false>, dim3(tiles), dim3(BLK), 0, s, ki, vi, ko, vo, n, RADIX_BITS * p, b.hist + p * RADIX, st, tiles, st_bytes, gsh, ticket, b.tile_counter + p, b.err_flag);
    ki = ko; vi = vo;
    ko = (ko == b.keys_a) ? b.keys_b : b.keys_a;
    vo = (vo == b.vals_a) ? b.vals_b : b.vals_a;
  }
}

// capacity of one of the 64 lists of block-crossing nodes: every node of every block that appends to it
uint32_t rt_cross_capacity(uint32_t n_tri) { const uint32_t nb = (n_tri + BLK - 1) / BLK; return ((nb + CROSS_SHARDS - 1) / CROSS_SHARDS) * BLK; }
uint32_t rt_build_counter_words() { return 32u + CROSS_SHARDS * CROSS_STRIDE; }   // sort tickets | shard counters of the crossing-node lists

// 16-leaf group boxes | 64-leaf chunk boxes | sparse tables of block boxes | super-block boxes | every block's first key
size_t rt_refit_table_float4s(uint32_t n_tri) {
  const size_t nb = ((size_t)n_tri + BLK - 1) / BLK, ns = (nb + SUPER - 1) / SUPER;
  return 2 * nb * (BLK / 16) + 2 * nb * (BLK / 64) + 2 * ns * ST_LEVELS * SUPER + 2 * ns + (nb + 3) / 4;
}

namespace {
struct RefitTables { float4 *mini_box, *chunk_box, *st, *super_box; uint32_t* block_key; uint32_t nb, ns; };
RefitTables refit_tables(const RtBuildBuffers& b) {
  RefitTables t;
  t.nb = (b.n_tri + BLK - 1) / BLK; t.ns = (t.nb + SUPER - 1) / SUPER;
  t.mini_box = b.refit_tables;
  t.chunk_box = t.mini_box + 2 * (size_t)t.nb * (BLK / 16);
  t.st = t.chunk_box + 2 * (size_t)t.nb * (BLK / 64);
  t.super_box = t.st + 2 * (size_t)t.ns * ST_LEVELS * SUPER;
  t.block_key = reinterpret_cast<uint32_t*>(t.super_box + 2 * (size_t)t.ns);
  return t;
}
}  // namespace

// leaves, triangle records, the tree and boxes inside every 256-leaf block (keys_a / vals_a = the sorted keys / order)
void rt_launch_refit_local(const RtBuildBuffers& b, float* nodes, float* tri_pos, float* tri_nor, float4* wide, float4* tri_pre, bool v2,
                           hipStream_t s) {
  const RefitTables t = refit_tables(b);
  hipLaunchKernelGGL(k_refit_local, dim3(t.nb), dim3(BLK), 0, s, b.mesh_rec, b.vals_a, b.keys_a, b.n_tri, b.span,
                     nodes, tri_pos, tri_nor, wide, tri_pre, b.texcoords ? b.tri_uv : nullptr, t.mini_box, t.chunk_box,
                     t.block_key, b.tile_counter + 32, b.cross_list, rt_cross_capacity(b.n_tri), v2);
}

// the nodes above the blocks: sparse tables of block boxes, then tree search + range queries for every crossing node
void rt_launch_refit_upper(const RtBuildBuffers& b, float* nodes, float4* wide, bool v2, hipStream_t s) {
  if (b.n_tri <= BLK) return;                        // one block: every node is block-local
  const RefitTables t = refit_tables(b);
  const uint32_t* cnt = b.tile_counter + 32;
  const uint32_t cap = rt_cross_capacity(b.n_tri);
  // sparse tables of block boxes || tree search of the crossing nodes; then their boxes (both list-driven: fixed grids)
  const uint32_t search_blocks = 2048;
  hipLaunchKernelGGL(k_refit_mid, dim3(t.ns + search_blocks), dim3(SUPER), 0, s, t.chunk_box, (b.n_tri + 63u) / 64u, t.nb, t.ns, t.st, t.super_box,
                     b.keys_a, t.block_key, b.n_tri, cnt, b.cross_list, cap, b.span, b.split);
  hipLaunchKernelGGL(k_refit_spans, dim3(2048), dim3(BLK), 0, s, cnt, b.cross_list, cap, b.span, b.split, b.n_tri, nodes, wide,
                     t.mini_box, t.chunk_box, t.st, t.super_box, v2);
}

// ---------------------------------------------------------------- paired traversal table (experiment: RTCORE_WIDE_PAIRED)
// Two tree levels per 128-byte line: a divergent fetch of a 64-byte record moves its whole line into the L1 anyway, so the
// table is laid out as PAIRS {head, tail} = a node and that child of it a ray is most likely to enter (the internal child with
// the larger surface).  Nothing but the records' positions changes (child references are translated to positions; leaf
// references are untouched), so the traversal code, its results and its counters are the ones of `wide`.
// A node is a tail iff the chain of "chosen" links that ends at it has odd length; every other internal node is a head.
RT_DEV float half_area(float4 mn, float4 mx) {
  const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
  return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(BLK) void k_pair_links(const float4* __restrict__ wide, uint32_t n_int, uint32_t* __restrict__ up) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_int) return;
  const float4 q0 = wide[4 * (size_t)i + 0], q1 = wide[4 * (size_t)i + 1], q2 = wide[4 * (size_t)i + 2], q3 = wide[4 * (size_t)i + 3];
  const uint32_t r0 = __float_as_uint(q0.w), r1 = __float_as_uint(q1.w);
  const bool i0 = !(r0 & 0x80000000u), i1 = !(r1 & 0x80000000u);
  const bool pick0 = i0 && (!i1 || half_area(q0, q1) >= half_area(q2, q3));
  if (i0) up[r0] = (i << 1) | (pick0 ? 1u : 0u);
  if (i1) up[r1] = (i << 1) | ((!pick0) ? 1u : 0u);
  if (i == 0) up[0] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(BLK) void k_pair_heads(const uint32_t* __restrict__ up, uint32_t n_int, uint32_t* __restrict__ pos, uint32_t* counter) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  bool head = false;
  if (i < n_int) {
    uint32_t j = i, len = 0;
    while (j != 0 && (up[j] & 1u)) { len += 1; j = up[j] >> 1; }
    head = (len & 1u) == 0u;
  }
  const unsigned long long m = __ballot(head && i != 0);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, 0);
  if (i < n_int) {
    if (i == 0) pos[0] = 0u;                                       // the root's pair is the table's first line
    else if (head) pos[i] = 2u * (1u + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
    else pos[i] = 0xFFFFFFFFu;
  }
}
__global__ __launch_bounds__(BLK) void k_pair_tails(const uint32_t* __restrict__ up, uint32_t n_int, uint32_t* __restrict__ pos) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_int || i == 0) return;
  if (pos[i] == 0xFFFFFFFFu) pos[i] = pos[up[i] >> 1] + 1u;        // (its parent is a head: k_pair_heads, the launch before)
}
__global__ __launch_bounds__(BLK) void k_pair_permute(const float4* __restrict__ wide, uint32_t n_int, const uint32_t* __restrict__ pos, float4* __restrict__ paired) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_int) return;
  float4 q0 = wide[4 * (size_t)i + 0], q1 = wide[4 * (size_t)i + 1];
  const float4 q2 = wide[4 * (size_t)i + 2], q3 = wide[4 * (size_t)i + 3];
  const uint32_t r0 = __float_as_uint(q0.w), r1 = __float_as_uint(q1.w);
  if (!(r0 & 0x80000000u)) q0.w = __uint_as_float(pos[r0]);
  if (!(r1 & 0x80000000u)) q1.w = __uint_as_float(pos[r1]);
  float4* o = paired + 4 * (size_t)pos[i];
  o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
}
// up / pos: n_int words each; counter: one zeroed word; paired: 2 * n_int records (one pair per head at most)
void rt_launch_pair_wide(const float4* wide, uint32_t n_int, uint32_t* up, uint32_t* pos, uint32_t* counter, float4* paired, hipStream_t s) {
  if (n_int == 0) return;
  const dim3 g((n_int + BLK - 1) / BLK), b(BLK);
  hipLaunchKernelGGL(k_pair_links, g, b, 0, s, wide, n_int, up);
  hipLaunchKernelGGL(k_pair_heads, g, b, 0, s, up, n_int, pos, counter);
  hipLaunchKernelGGL(k_pair_tails, g, b, 0, s, up, n_int, pos);
  hipLaunchKernelGGL(k_pair_permute, g, b, 0, s, wide, n_int, pos, paired);
}

// root_ref / err words live right behind the bounds words: see rt_api.hip (scratch[0], scratch[1])
void rt_launch_pack_impl(const RtBvhDev& bvh, uint32_t* d_root_ref, uint32_t* d_err, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_wide, dim3((bvh.n_nodes + BLK - 1) / BLK), dim3(BLK), 0, s, bvh.nodes, bvh.n_nodes, bvh.wide,
                     bvh.n_wide, bvh.leaf_shift, bvh.v2, bvh.n_tri, d_root_ref, d_err);
  hipLaunchKernelGGL(k_pack_tris, dim3((bvh.n_tri + BLK - 1) / BLK), dim3(BLK), 0, s, bvh.tri_pos, bvh.n_tri, bvh.tri_pre);
}
