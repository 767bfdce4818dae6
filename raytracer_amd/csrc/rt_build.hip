// rt_build.hip — hot path 1: Morton LBVH build on gfx950 (wave64).
//
// Pipeline (the reference only has the Morton shader,
// web/morton_compute.wgsl, and never dispatches it — F4):
//   k_bounds*      centroid min/max reduction (order-independent => bit-exact)
//   k_morton       morton_compute.wgsl:33-69, one thread per triangle
//   k_sort_*       onesweep LSD radix sort, 4 x 8-bit digits, (key, payload) pairs, STABLE:
//                  one global histogram pass, then per digit a single pass with
//                  chained-scan decoupled look-back, wave64 ballot/popcount ranking and an
//                  LDS-staged scatter
//   k_karras       Karras 2012 radix tree, one thread per internal node, ties broken by index
//   k_refit        one thread per leaf: emit BVH-ordered triangle buffers + leaf record, then
//                  climb; the second arrival at a node (atomic counter) unions the children
//   k_pack_*       derive the traversal layout (64-byte two-child records, hoisted edges)
// All kernels are HBM-bound integer/min-max work: coalesced 4-16 B/lane accesses, LDS staging,
// no MFMA.
#include "rt_device.h"
#include "rt_internal.h"

namespace {

constexpr int BLK = 256;
constexpr int RADIX = 256;
constexpr int SORT_KPT = 16;                       // keys per thread
constexpr int SORT_TILE = BLK * SORT_KPT;          // 4096 keys per tile
constexpr uint32_t ST_AGG = 1u << 30, ST_INCL = 2u << 30, ST_MASK = (1u << 30) - 1u;
constexpr uint32_t SPIN_LIMIT = 1u << 24;          // bounded look-back spin (error flag instead of a hang)

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

// ---------------------------------------------------------------- centroid bounds
// Two stages, no atomics: every block leaves its six ordered-uint partials in its own slot (2048 blocks x 6 same-address
// atomicMin/Max serialised at ~26 ns each: 53 of the kernel's 62 us at 1.3 M triangles), one block folds the slots.
// Also keeps every centroid (12 B/triangle, coalesced) so that k_morton streams them instead of repeating the
// three-vertex gather through the index buffer (48 B/triangle of scattered reads).
__global__ __launch_bounds__(BLK) void k_bounds(const float* __restrict__ pos, const uint32_t* __restrict__ vidx,
                                                uint32_t n_tri, uint32_t* __restrict__ part /* [gridDim.x][6] */,
                                                float* __restrict__ centroids /* [3T] */) {
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (uint32_t t = blockIdx.x * BLK + threadIdx.x; t < n_tri; t += gridDim.x * BLK) {
    V3 c = tri_centroid(pos, vidx, t);
    centroids[3 * (size_t)t + 0] = c.x; centroids[3 * (size_t)t + 1] = c.y; centroids[3 * (size_t)t + 2] = c.z;
    uint32_t e[3] = {f2ord(c.x), f2ord(c.y), f2ord(c.z)};
#pragma unroll
    for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], e[k]); mx[k] = max(mx[k], e[k]); }
  }
  __shared__ uint32_t s[6][BLK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    part[(size_t)blockIdx.x * 6 + k] = v;
  }
}

// fold the block partials (6 waves, one per component), then the `Bounds` uniform of morton_compute.wgsl:3-6:
// invExtent = 1/(max-min), 0 where the extent is 0
__global__ __launch_bounds__(384) void k_bounds_final(const uint32_t* __restrict__ part, uint32_t n_blocks, float* bounds_f,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ tile_counter) {
  __shared__ uint32_t s[6];
  // the sort's global digit histograms and tile tickets start at zero (saves three memset launches)
  for (int i = threadIdx.x; i < 4 * RADIX; i += 384) hist[i] = 0u;
  if (threadIdx.x < 4) tile_counter[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  uint32_t v = (k < 3) ? 0xFFFFFFFFu : 0u;
  for (uint32_t b = lane; b < n_blocks; b += 64) {
    const uint32_t x = part[(size_t)b * 6 + k];
    v = (k < 3) ? min(v, x) : max(v, x);
  }
  v = (k < 3) ? wave_min_u32(v) : wave_max_u32(v);
  if (lane == 0) s[k] = v;
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    float lo = ord2f(s[c]), hi = ord2f(s[3 + c]);
    float ext = hi - lo;
    bounds_f[c] = lo;
    bounds_f[4 + c] = (ext > 0.0f) ? (1.0f / ext) : 0.0f;
  } else if (threadIdx.x == 3) { bounds_f[3] = 0.0f; bounds_f[7] = 0.0f; }
}

// ---------------------------------------------------------------- Morton codes
__global__ __launch_bounds__(BLK) void k_morton(const float* __restrict__ centroids,
                                                uint32_t n_tri, const float* __restrict__ bounds_f,
                                                uint32_t* __restrict__ codes, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, uint32_t* __restrict__ tile_status, size_t n_status) {
  // the look-back words of the four sort passes start at zero (their own memset launch otherwise)
  for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n_status; i += (size_t)gridDim.x * BLK) tile_status[i] = 0u;
  const V3 bmin = v3(bounds_f[0], bounds_f[1], bounds_f[2]), inv = v3(bounds_f[4], bounds_f[5], bounds_f[6]);
  for (uint32_t t = blockIdx.x * BLK + threadIdx.x; t < n_tri; t += gridDim.x * BLK) {
    const V3 ctr = v3(centroids[3 * (size_t)t + 0], centroids[3 * (size_t)t + 1], centroids[3 * (size_t)t + 2]);   // :52, from k_bounds
    V3 nrm = (ctr - bmin) * inv;                                          // :55
    uint32_t code = (expand_bits(quantize10(nrm.x)) << 2) | (expand_bits(quantize10(nrm.y)) << 1) |
                    expand_bits(quantize10(nrm.z));                      // :58-68
    codes[t] = code; keys[t] = code; vals[t] = t;
  }
}

// ---------------------------------------------------------------- onesweep radix sort
__global__ __launch_bounds__(BLK) void k_sort_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[4][RADIX];
  for (int i = threadIdx.x; i < 4 * RADIX; i += BLK) (&h[0][0])[i] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
    uint32_t k = keys[i];
#pragma unroll
    for (int p = 0; p < 4; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * RADIX; i += BLK) {
    uint32_t v = (&h[0][0])[i];
    if (v) atomicAdd(&hist[i], v);
  }
}

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

__global__ __launch_bounds__(BLK) void k_sort_scan(uint32_t* hist) {   // 4 blocks: one per digit position
  __shared__ uint32_t s_wave[BLK / 64];
  uint32_t* h = hist + blockIdx.x * RADIX;
  uint32_t v = h[threadIdx.x];
  uint32_t ex = block_excl_scan(v, s_wave, nullptr);
  h[threadIdx.x] = ex;
}

__global__ __launch_bounds__(BLK) void k_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                  uint32_t n, int shift, const uint32_t* __restrict__ digit_base,
                                                  uint32_t* status, uint32_t* tile_counter, uint32_t* err_flag) {
  __shared__ uint32_t s_keys[SORT_TILE];
  __shared__ uint32_t s_vals[SORT_TILE];
  __shared__ uint32_t s_whist[BLK / 64][RADIX];
  __shared__ uint32_t s_dstart[RADIX];
  __shared__ uint32_t s_gbase[RADIX];
  __shared__ uint32_t s_wave[BLK / 64];
  __shared__ uint32_t s_tile;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tile ids are handed out in start order, so every predecessor tile is already running:
  // the look-back below cannot wait on a block that has not been scheduled.
  if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
  for (int i = tid; i < (BLK / 64) * RADIX; i += BLK) (&s_whist[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t base = tile * (uint32_t)SORT_TILE;
  const uint32_t wbase = base + (uint32_t)wave * (64 * SORT_KPT);

  uint32_t key[SORT_KPT], val[SORT_KPT], rank[SORT_KPT];
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    val[r] = valid ? vals_in[idx] : 0u;
  }
  // wave-level stable ranking: peers = lanes holding the same digit (8 ballots), rank inside
  // the peer group by popcount below the lane; the group leader bumps the wave's digit counter.
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    const bool valid = (wbase + r * 64 + lane) < n;
    const uint32_t d = (key[r] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    int leader = lane;
    uint32_t prev = 0;
    if (valid) {
      leader = __ffsll((long long)peers) - 1;
      if (lane == leader) prev = atomicAdd(&s_whist[wave][d], (uint32_t)__popcll(peers));
    }
    prev = (uint32_t)__shfl((int)prev, leader);
    rank[r] = prev + (uint32_t)__popcll(peers & lt_mask);
  }
  __syncthreads();

  // thread d owns digit d: per-wave exclusive offsets, tile total, look-back
  uint32_t total = 0;
  {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) { uint32_t c = s_whist[w][tid]; s_whist[w][tid] = run; run += c; }
    total = run;
  }
  uint32_t* my_status = status + (size_t)tile * RADIX + tid;
  __hip_atomic_store(my_status, (tile == 0 ? ST_INCL : ST_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t dstart = block_excl_scan(total, s_wave, nullptr);
  uint32_t excl = 0;
  if (tile > 0) {
    uint32_t p = tile - 1, spins = 0;
    for (;;) {
      const uint32_t sw = __hip_atomic_load(status + (size_t)p * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t f = sw >> 30;
      if (f == 0) {
        if (++spins > SPIN_LIMIT) { atomicOr(err_flag, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      excl += sw & ST_MASK;
      if (f == 2 || p == 0) break;
      --p;
    }
    __hip_atomic_store(my_status, ST_INCL | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s_dstart[tid] = dstart;
  s_gbase[tid] = digit_base[tid] + excl - dstart;
  __syncthreads();

  // LDS-staged scatter: tile-local sorted order first, then coalesced runs to HBM
#pragma unroll
  for (int r = 0; r < SORT_KPT; ++r) {
    if ((wbase + r * 64 + lane) < n) {
      const uint32_t d = (key[r] >> shift) & 255u;
      const uint32_t pos = s_dstart[d] + s_whist[wave][d] + rank[r];
      s_keys[pos] = key[r];
      s_vals[pos] = val[r];
    }
  }
  __syncthreads();
  const uint32_t cnt = (n - base < (uint32_t)SORT_TILE) ? (n - base) : (uint32_t)SORT_TILE;
#pragma unroll
  for (int k = 0; k < SORT_KPT; ++k) {
    const uint32_t i = tid + k * BLK;
    if (i < cnt) {
      const uint32_t kk = s_keys[i];
      const uint32_t out = s_gbase[(kk >> shift) & 255u] + i;
      keys_out[out] = kk;
      vals_out[out] = s_vals[i];
    }
  }
}

// ---------------------------------------------------------------- Karras radix tree
// delta(i,j): common prefix of the augmented keys (code, index); -1 when j is out of range
RT_DEV int kdelta(const uint32_t* __restrict__ k, int n, int i, uint32_t ki, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t kj = k[j];
  if (ki != kj) return __clz((int)(ki ^ kj));
  return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

__global__ __launch_bounds__(BLK) void k_karras(const uint32_t* __restrict__ k, uint32_t n_tri,
                                                int32_t* __restrict__ child, int32_t* __restrict__ parent,
                                                uint32_t* __restrict__ visit) {
  const int n = (int)n_tri;
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i == 0) parent[0] = -1;
  if (i >= n - 1) return;
  visit[i] = 0u;                                   // the refit's arrival counter of this node (its own memset launch otherwise)
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
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  // node ids: internal k -> k, leaf k -> (T-1)+k
  const int c0 = (lo == gamma) ? (n - 1) + gamma : gamma;
  const int c1 = (hi == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
  child[2 * (size_t)i + 0] = c0; child[2 * (size_t)i + 1] = c1;
  // bit 30: this node's leaf range [lo, hi] lies inside one 256-leaf block of k_refit (block-local climb)
  const int link = i | (((lo >> 8) == (hi >> 8)) ? (1 << 30) : 0);
  parent[c0] = link; parent[c1] = link;
}

// ---------------------------------------------------------------- leaves + bottom-up refit
// Node records are handed from the first arrival at a parent to the second one, which may run
// on another CU / XCD (L1 per CU, L2 per XCD are not coherent).  Two forms:
//   FENCED = false (default): every record is written with 8-byte agent-scope (sc1, write-
//     through) stores, drained with s_waitcnt vmcnt(0) before the arrival counter is bumped;
//     the second arrival reads the sibling with 8-byte agent-scope (sc1) loads.  No L2
//     write-back / L1 invalidate per hand-off (MI355X_MICROARCH.md "Valid forms", row 1).
//     k_refit_verify re-derives every internal box from its children afterwards; on any
//     mismatch the host re-runs the FENCED form.
//   FENCED = true: plain stores + __threadfence() (release / acquire at agent scope) around
//     the counter: the textbook form, ~80x slower here (buffer_wbl2 + buffer_inv per arrival).
RT_DEV unsigned long long pack2f(float a, float b) {
  return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
// through == false: the record is only read by later kernels (its parent is block-local, or it is the root): plain stores,
// which L2 merges into full lines instead of four 8-byte write-throughs.
template <bool FENCED>
RT_DEV void store_node(float* nodes, int idx, V3 mn, float a, V3 mx, float b, bool through = true) {
  if (FENCED || !through) {
    float4* nd = reinterpret_cast<float4*>(nodes) + 2 * (size_t)idx;
    nd[0] = make_float4(mn.x, mn.y, mn.z, a);
    nd[1] = make_float4(mx.x, mx.y, mx.z, b);
  } else {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(nodes) + 4 * (size_t)idx;
    __hip_atomic_store(q + 0, pack2f(mn.x, mn.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, pack2f(mn.z, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, pack2f(mx.x, mx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 3, pack2f(mx.z, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool FENCED>
RT_DEV void load_box(const float* nodes, int idx, V3& mn, V3& mx) {
  if (FENCED) {
    const float4* nd = reinterpret_cast<const float4*>(nodes) + 2 * (size_t)idx;
    const float4 s0 = nd[0], s1 = nd[1];
    mn = xyz(s0); mx = xyz(s1);
  } else {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(nodes) + 4 * (size_t)idx;
    const unsigned long long w0 = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w2 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w3 = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mn = v3(__uint_as_float((uint32_t)w0), __uint_as_float((uint32_t)(w0 >> 32)), __uint_as_float((uint32_t)w1));
    mx = v3(__uint_as_float((uint32_t)w2), __uint_as_float((uint32_t)(w2 >> 32)), __uint_as_float((uint32_t)w3));
  }
}

// Block-local climb.  Block b owns sorted leaves [256 b, 256 b + 255].  k_karras marks a parent link with PARENT_LOCAL
// when the parent's whole leaf range lies inside one such block; since an internal node's index is one end of its own
// range, both the node and its two children then belong to that block.  For those nodes the hand-off (arrival counter +
// sibling box) goes through LDS — no global atomic, no write-through wait — and only the nodes whose range crosses a block
// boundary (about 1 in 128) take the global path described above.  Every record is still written to HBM (interchange
// `nodes`, traversal `wide`), the global path reads a locally finished sibling from there.
constexpr int PARENT_LOCAL = 1 << 30;
constexpr int PARENT_MASK = PARENT_LOCAL - 1;
static_assert(BLK == 256, "k_karras marks block-local nodes with (lo >> 8) == (hi >> 8)");

template <bool FENCED>
__global__ __launch_bounds__(BLK) void k_refit(const float* __restrict__ pos, const float* __restrict__ nor,
                                               const uint32_t* __restrict__ vidx, const uint32_t* __restrict__ nidx,
                                               const uint32_t* __restrict__ order, uint32_t n_tri,
                                               const int32_t* __restrict__ child, const int32_t* __restrict__ parent,
                                               uint32_t* visit, float* nodes, float* __restrict__ tri_pos,
                                               float* __restrict__ tri_nor, float4* __restrict__ wide, float4* __restrict__ tri_pre,
                                               const float* __restrict__ texcoords, const uint32_t* __restrict__ tex_idx,
                                               const int32_t* __restrict__ material_ids, float4* __restrict__ tri_uv,
                                               int leaf_shift, bool v2) {
  __shared__ float s_box[2 * BLK][6];      // [0, BLK): this block's leaves, [BLK, 2 BLK): internal nodes with index in the block's range
  __shared__ uint32_t s_visit[BLK];        // arrival counters of the block-local internal nodes
  __shared__ int s_child[BLK][2];          // links of the internal nodes with index in the block's range, read once, coalesced:
  __shared__ int s_parent[BLK];            // the local climb then has no dependent global load per level
  s_visit[threadIdx.x] = 0u;
  const uint32_t j = blockIdx.x * BLK + threadIdx.x;
  if (j + 1 < n_tri) {
    const int2 c = reinterpret_cast<const int2*>(child)[j];
    s_child[threadIdx.x][0] = c.x; s_child[threadIdx.x][1] = c.y;
    s_parent[threadIdx.x] = parent[j];
  }
  __syncthreads();
  if (j >= n_tri) return;
  const int first_leaf = (int)n_tri - 1;   // leaf j is node (T-1)+j, one triangle, start = j
  const uint32_t ti = order[j];
  V3 p[3];
  float4* tp = reinterpret_cast<float4*>(tri_pos) + (size_t)j * 3;
  float4* tn = reinterpret_cast<float4*>(tri_nor) + (size_t)j * 3;
#pragma unroll
  for (int v = 0; v < 3; ++v) {                     // web/script.js:530-545
    p[v] = load_vtx(pos, vidx[3 * (size_t)ti + v]);
    V3 nn = load_vtx(nor, nidx[3 * (size_t)ti + v]);
    tp[v] = make_float4(p[v].x, p[v].y, p[v].z, 0.0f);
    tn[v] = make_float4(nn.x, nn.y, nn.z, 0.0f);
  }
  {                                                  // traversal layout of the triangle, the statements of k_pack_tris
    const V3 e1 = p[1] - p[0], e2 = p[2] - p[0];
    const V3 ng = cross3(e1, e2);
    tri_pre[3 * (size_t)j + 0] = make_float4(p[0].x, p[0].y, p[0].z, ng.x);
    tri_pre[3 * (size_t)j + 1] = make_float4(e1.x, e1.y, e1.z, ng.y);
    tri_pre[3 * (size_t)j + 2] = make_float4(e2.x, e2.y, e2.z, ng.z);
  }
  if (texcoords) {                                   // surface record (rtcore.h "material / texture shading"), as k_gather_uv
    const float2* tc = reinterpret_cast<const float2*>(texcoords);
    const float2 a = tc[tex_idx[3 * (size_t)ti + 0]], b = tc[tex_idx[3 * (size_t)ti + 1]], c = tc[tex_idx[3 * (size_t)ti + 2]];
    const int32_t m = material_ids ? material_ids[ti] : -1;
    tri_uv[2 * (size_t)j + 0] = make_float4(a.x, a.y, b.x, b.y);
    tri_uv[2 * (size_t)j + 1] = make_float4(c.x, c.y, __uint_as_float((uint32_t)m), 0.0f);
  }
  // true triangle bounds (web/script.js:309-316)
  V3 mn = v3(fminf(fminf(p[0].x, p[1].x), p[2].x), fminf(fminf(p[0].y, p[1].y), p[2].y), fminf(fminf(p[0].z, p[1].z), p[2].z));
  V3 mx = v3(fmaxf(fmaxf(p[0].x, p[1].x), p[2].x), fmaxf(fmaxf(p[0].y, p[1].y), p[2].y), fmaxf(fmaxf(p[0].z, p[1].z), p[2].z));
  int cur = first_leaf + (int)j;
  int cur_slot = (int)threadIdx.x;                   // LDS slot of `cur`
  int pe = parent[cur];                              // link to the parent of `cur`
  store_node<FENCED>(nodes, cur, mn, enc_index(v2, j), mx, enc_leaf_b(v2, 1u),   // A = tri_start, B = -count (script.js:348,554,558)
                     pe >= 0 && !(pe & PARENT_LOCAL));
  for (;;) {
    if (pe < 0) break;
    const bool local = (pe & PARENT_LOCAL) != 0;
    const int par = pe & PARENT_MASK;
    uint32_t arrived;
    int gc0 = 0, gc1 = 0, gpe = -1;
    if (!local) {                                      // the global path's links do not depend on who arrives second: issue the loads
      const int2 c = reinterpret_cast<const int2*>(child)[par];   // before the wait + atomic instead of two more round trips after them
      gc0 = c.x; gc1 = c.y; gpe = parent[par];
    }
    if (local) {
      float* sb = s_box[cur_slot];
      sb[0] = mn.x; sb[1] = mn.y; sb[2] = mn.z; sb[3] = mx.x; sb[4] = mx.y; sb[5] = mx.z;
      __threadfence_block();                           // the box is in LDS before the counter moves (one LDS, in-order)
      arrived = atomicAdd(&s_visit[par & (BLK - 1)], 1u);
    } else {
      if (FENCED) __threadfence();                     // release this subtree's records (agent scope)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have left this CU
      arrived = atomicAdd(&visit[par], 1u);
    }
    if (arrived == 0) break;                           // the sibling's thread will finish this node
    const int c0 = local ? s_child[par & (BLK - 1)][0] : gc0;
    const int c1 = local ? s_child[par & (BLK - 1)][1] : gc1;
    const int sib = (c0 == cur) ? c1 : c0;
    V3 smn, smx;
    if (local) {
      __threadfence_block();
      const int sslot = (sib >= first_leaf) ? ((sib - first_leaf) & (BLK - 1)) : (BLK + (sib & (BLK - 1)));
      const float* sb = s_box[sslot];
      smn = v3(sb[0], sb[1], sb[2]); smx = v3(sb[3], sb[4], sb[5]);
    } else {
      if (FENCED) __threadfence();                     // acquire the sibling's record
      load_box<FENCED>(nodes, sib, smn, smx);
    }
    {                                                  // 64-byte two-child record of `par`, the statements of k_pack_wide
      const bool cur0 = (c0 == cur);
      const V3 a_mn = cur0 ? mn : smn, a_mx = cur0 ? mx : smx, b_mn = cur0 ? smn : mn, b_mx = cur0 ? smx : mx;
      const uint32_t r0 = (c0 >= first_leaf) ? (0x80000000u | (uint32_t)(c0 - first_leaf)) : (uint32_t)c0;
      const uint32_t r1 = (c1 >= first_leaf) ? (0x80000000u | (uint32_t)(c1 - first_leaf)) : (uint32_t)c1;
      wide[4 * (size_t)par + 0] = make_float4(a_mn.x, a_mn.y, a_mn.z, __uint_as_float(r0));
      wide[4 * (size_t)par + 1] = make_float4(a_mx.x, a_mx.y, a_mx.z, __uint_as_float(r1));
      wide[4 * (size_t)par + 2] = make_float4(b_mn.x, b_mn.y, b_mn.z, 0.0f);
      wide[4 * (size_t)par + 3] = make_float4(b_mx.x, b_mx.y, b_mx.z, 0.0f);
    }
    mn = v3(fminf(mn.x, smn.x), fminf(mn.y, smn.y), fminf(mn.z, smn.z));
    mx = v3(fmaxf(mx.x, smx.x), fmaxf(mx.y, smx.y), fmaxf(mx.z, smx.z));
    pe = local ? s_parent[par & (BLK - 1)] : gpe;
    store_node<FENCED>(nodes, par, mn, enc_index(v2, (uint32_t)c0), mx, enc_index(v2, (uint32_t)c1),   // v1: indices as f32 values (F11)
                       pe >= 0 && !(pe & PARENT_LOCAL));
    cur = par;
    cur_slot = BLK + (par & (BLK - 1));              // only meaningful while the climb stays local (ranges nest)
  }
}

// every internal box must equal the union of its children's boxes, every slot must be filled
__global__ __launch_bounds__(BLK) void k_refit_verify(const float* __restrict__ nodes, uint32_t n_tri,
                                                      const int32_t* __restrict__ child, const uint32_t* __restrict__ visit,
                                                      uint32_t* err, bool v2) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i + 1 >= n_tri) return;
  const float4* nd = reinterpret_cast<const float4*>(nodes);
  const int c0 = child[2 * (size_t)i + 0], c1 = child[2 * (size_t)i + 1];
  const float4 a = nd[2 * (size_t)i], b = nd[2 * (size_t)i + 1];
  const float4 a0 = nd[2 * (size_t)c0], b0 = nd[2 * (size_t)c0 + 1], a1 = nd[2 * (size_t)c1], b1 = nd[2 * (size_t)c1 + 1];
  bool ok = !dec_is_leaf(v2, b.w) && (dec_index(v2, a.w) == c0) && (dec_index(v2, b.w) == c1);
  ok = ok && a.x == fminf(a0.x, a1.x) && a.y == fminf(a0.y, a1.y) && a.z == fminf(a0.z, a1.z);
  ok = ok && b.x == fmaxf(b0.x, b1.x) && b.y == fmaxf(b0.y, b1.y) && b.z == fmaxf(b0.z, b1.z);
  if (!ok) atomicOr(err, 8u);
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

uint32_t rt_sort_tile_count(uint32_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
size_t rt_sort_status_words(uint32_t n) { return (size_t)4 * rt_sort_tile_count(n) * RADIX; }

uint32_t rt_bounds_words() { return 6u * 2048u; }   // grid_for's block cap x 6 partials
void rt_launch_bounds(const RtBuildBuffers& b, hipStream_t s) {
  const int g = grid_for(b.n_tri, BLK);
  hipLaunchKernelGGL(k_bounds, dim3(g), dim3(BLK), 0, s, b.pos, b.vidx, b.n_tri, b.bounds_enc, b.centroids);
  hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(384), 0, s, b.bounds_enc, (uint32_t)g, b.bounds_f, b.hist, b.tile_counter);
}

void rt_launch_morton(const RtBuildBuffers& b, hipStream_t s) {
  hipLaunchKernelGGL(k_morton, dim3(grid_for(b.n_tri, BLK)), dim3(BLK), 0, s, b.centroids, b.n_tri, b.bounds_f, b.codes,
                     b.keys_a, b.vals_a, b.tile_status, rt_sort_status_words(b.n_tri));
}

void rt_launch_sort(const RtBuildBuffers& b, hipStream_t s) {
  const uint32_t n = b.n_tri, tiles = rt_sort_tile_count(n);
  // hist / tile_counter were zeroed by k_bounds_final, tile_status by k_morton
  hipLaunchKernelGGL(k_sort_hist, dim3(grid_for(n, BLK * 8)), dim3(BLK), 0, s, b.keys_a, n, b.hist);
  hipLaunchKernelGGL(k_sort_scan, dim3(4), dim3(BLK), 0, s, b.hist);
  uint32_t *ki = b.keys_a, *vi = b.vals_a, *ko = b.keys_b, *vo = b.vals_b;
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(k_onesweep, dim3(tiles), dim3(BLK), 0, s, ki, vi, ko, vo, n, 8 * p, b.hist + p * RADIX,
                       b.tile_status + (size_t)p * tiles * RADIX, b.tile_counter + p, b.err_flag);
    uint32_t* t;
    t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
}

void rt_launch_karras(const RtBuildBuffers& b, hipStream_t s) {
  const uint32_t n_int = b.n_tri > 1 ? b.n_tri - 1 : 1;
  hipLaunchKernelGGL(k_karras, dim3((n_int + BLK - 1) / BLK), dim3(BLK), 0, s, b.keys_a, b.n_tri, b.child, b.parent, b.visit);
}

void rt_launch_refit(const RtBuildBuffers& b, float* nodes, float* tri_pos, float* tri_nor, float4* wide, float4* tri_pre,
                     int leaf_shift, bool fenced, bool v2, hipStream_t s) {
  const dim3 g((b.n_tri + BLK - 1) / BLK), blk(BLK);
  if (fenced)
    hipLaunchKernelGGL(k_refit<true>, g, blk, 0, s, b.pos, b.nor, b.vidx, b.nidx, b.vals_a, b.n_tri, b.child, b.parent, b.visit,
                       nodes, tri_pos, tri_nor, wide, tri_pre, b.texcoords, b.tex_idx, b.material_ids, b.tri_uv, leaf_shift, v2);
  else
    hipLaunchKernelGGL(k_refit<false>, g, blk, 0, s, b.pos, b.nor, b.vidx, b.nidx, b.vals_a, b.n_tri, b.child, b.parent, b.visit,
                       nodes, tri_pos, tri_nor, wide, tri_pre, b.texcoords, b.tex_idx, b.material_ids, b.tri_uv, leaf_shift, v2);
  if (b.n_tri > 1)   // sets bit 8 of err_flag on any inconsistency
    hipLaunchKernelGGL(k_refit_verify, dim3((b.n_tri - 1 + BLK - 1) / BLK), blk, 0, s, nodes, b.n_tri, b.child, b.visit, b.err_flag, v2);
}

// root_ref / err words live right behind the bounds words: see rt_api.hip (scratch[0], scratch[1])
void rt_launch_pack_impl(const RtBvhDev& bvh, uint32_t* d_root_ref, uint32_t* d_err, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_wide, dim3((bvh.n_nodes + BLK - 1) / BLK), dim3(BLK), 0, s, bvh.nodes, bvh.n_nodes, bvh.wide,
                     bvh.n_wide, bvh.leaf_shift, bvh.v2, bvh.n_tri, d_root_ref, d_err);
  hipLaunchKernelGGL(k_pack_tris, dim3((bvh.n_tri + BLK - 1) / BLK), dim3(BLK), 0, s, bvh.tri_pos, bvh.n_tri, bvh.tri_pre);
}
