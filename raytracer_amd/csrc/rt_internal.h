// rt_internal.h — shared declarations between the C-ABI layer (rt_api.hip) and the kernel
// translation units (rt_build.hip, rt_trace.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/rtcore.h"

// ---------------------------------------------------------------- device memory layout (HBM)
// Interchange (public, web/script.js:526-561):
//   nodes    f32[8N]   32 B/node  {mn.xyz, A | mx.xyz, B}   (A/B: v1 f32 values or v2 u32 bit-casts, rt_device.h)
//   tri_pos  f32[12T]  48 B/tri   3 x (x,y,z,0)
//   tri_nor  f32[12T]  48 B/tri
// Traversal layout (internal, derived by rt_pack_*):
//   wide     float4[4*Ni] 64 B per internal node i:
//              a = {c0.mn.xyz, bits(c0ref)}  b = {c0.mx.xyz, bits(c1ref)}
//              c = {c1.mn.xyz, 0}            d = {c1.mx.xyz, 0}
//            ref: bit31 clear = internal node index; bit31 set = leaf:
//                 start = ref & ((1<<leaf_shift)-1), count = ((ref & 0x7fffffff) >> leaf_shift) + 1
//   tri_pre  float4[3T]  48 B/tri: {v0.xyz, ng.x} {e1.xyz, ng.y} {e2.xyz, ng.z}
//            e1 = v1-v0, e2 = v2-v0, ng = cross(e1,e2): the exact sub-expressions of
//            triIntersect_fast / the back-face pre-cull (compute.wgsl:114-115,282), hoisted.
struct RtBvhDev {
  float*  nodes = nullptr;
  float*  tri_pos = nullptr;
  float*  tri_nor = nullptr;
  float4* wide = nullptr;
  float4* tri_pre = nullptr;
  uint32_t n_nodes = 0, n_tri = 0, n_wide = 0;
  uint32_t root_ref = 0;     // traversal ref of node 0
  int      leaf_shift = 23;
  bool     v2 = false;       // node record encoding of `nodes` (RT_NODES_V2)
  bool     owns_interchange = true;
  float    root_box[8] = {};   // host copy of node 0 (mn.xyz, A | mx.xyz, B) for the screen-space cull rectangle
};

// Surface extension (include/rtcore.h "material / texture shading")
struct RtTexDev { const uint32_t* texels; uint32_t w, h; };   // texel = R | G<<8 | B<<16 | 255<<24, row 0 = top

// internal third value of the LAYOUT template parameter: RT_LAYOUT_WIDE64 records addressed with 32-bit offsets (RtRenderParams::near32)
#define RT_LAYOUT_WIDE64_NEAR 2

struct RtRenderParams {
  const float4*   tri_uv;          // [2T] {u0 v0 u1 v1} {u2 v2 bits(material) 0}, BVH order; NULL unless RT_FLAG_TEXTURED
  const float4*   materials;       // [n_materials] {Kd.rgb, bits(texture slot)}
  const RtTexDev* textures;        // [n_textures]
  uint32_t n_materials, n_textures;
  const float*  nodes;
  const float*  tri_pos;
  const float*  tri_nor;
  const float4* wide;
  const float4* tri_pre;
  // (kernels: template LAYOUT = RT_LAYOUT_WIDE64_NEAR)
  // both traversal tables inside ONE 4 GB window (near32 != 0): records are then addressed as near_base + a 32-bit byte offset,
  // which the vector memory instructions take as scalar base + 32-bit lane offset; else 64-bit pointer arithmetic per fetch
  const char* near_base; uint32_t near_wide_off, near_pre_off, near32;
  uint32_t root_ref;
  int      leaf_shift;
  int      nodes_v2;               // encoding of `nodes` (read by the AOS32 walk only)
  uint32_t step_budget;            // n_nodes + n_tri + 64: loop iterations no valid tree needs more of
  float    U[44];
  int mode, gi_samples, gi_bounces, soft_samples, max_stack;
  uint32_t width, height;
  int stripe_rows, stripe_rank, stripe_world;   // stripe_rows == 0: whole image
  uint32_t tiles_x, n_tiles, local_rows;
  // inclusive pixel rectangle (global image coordinates) that conservatively contains every pixel whose
  // primary ray can enter the root box; tiles outside it are final without generating rays (x1 < x0: empty)
  int cull_x0, cull_x1, cull_y0, cull_y1;
  int frag;                        // 0 = compute.wgsl; 1 = web/shader.wgsl, 2 = shader.wgsl (RT_MODE_FRAG_*)
  uint32_t miss_rgba;              // pixel of a ray that hits nothing: (0,0,0,1), or the older shaders' sky
  uint8_t*  rgba8;     // packed local rows
  uint32_t* hit_tri;
  float*    hit_t;
  uint32_t* shard_heads;           // [rt_shard_words()] phase-B queue heads, one per 128-byte line; zeroed before launch
  uint32_t* shade_heads;           // [rt_shard_words()] the same for the shade launch of the traverse / shade split (NULL: no split)
  uint32_t* hits;                  // [list entries (x frames of a batch, list_stride apart)] traverse / shade split: triangle of the entry's primary hit, 0xFFFFFFFF = miss; NULL = whole pixel in one launch
  uint32_t shard_stride;           // words between shard heads
  uint32_t shard_group;            // consecutive chunks dealt to one shard at a time (round-robin over the shards)
  unsigned long long* masks;       // [3 * n_tiles] per tile and cost class: ballot of pixels whose primary ray enters the root box
  uint32_t* block_total;           // [3 * rt_cull_blocks(n_tiles)] per block and cost class
  uint32_t* list;                  // [local_rows*width] compacted pixel ids (phase A -> phase B): hot class first
  uint32_t* list_count;            // written by k_compact
  // Frame-coherent scheduling hints (never results): cost class of every pixel as measured by the previous frame
  // (0 = cheap or unknown, 1 = normal, 2 .. N-1 = hotter and hotter), read by k_cull, rewritten by k_render.  NULL = no hints.
  uint8_t*  pix_class;             // [local_rows*width]
  uint32_t  class_promote;         // k_cull: a tile with a hot pixel traces all its non-cheap pixels with the hot class
  uint32_t  class_t1, class_t2;    // traversal steps: class = (steps >= t1) + (steps >= t2); t2 == 0: adaptive, see class_hist
  uint32_t* class_hist;            // [64] histogram of (steps - t1) / 4 over last frame's pixels with steps >= t1 (k_render adds, k_compact reads + clears)
  uint32_t  class_hot_pct;         // adaptive t2: the hot class is about this share of last frame's non-cheap pixels (k_compact -> list_count[2])
  uint32_t  class_share[6];        // classes 2 .. N-1: class c holds about the top class_share[c-2] % of last frame's non-cheap pixels (descending; [0] = class_hot_pct)
  // PT split (sample-parallel GI, rt_trace.hip): phase 1 (k_render) leaves one record per hit pixel in the sub-list of the queue
  // shard its chunk came from; phase 2 (k_pt_samples) traces ONE GI sample of 64 records per wave item, phase 3 (k_pt_resolve) sums.  pt_cap == 0: no split.
  float4*   pt_recs;               // [N_SHARD][pt_cap][3]: {hit_pos, d0} {n_os, bits(pixel idx)} {albedo, bits(pixelIndex)}
  uint32_t* pt_count;              // [N_SHARD] records per sub-list           (zeroed by k_cull)
  uint32_t* pt_heads;              // [N_SHARD * shard_stride] phase-2 queue heads (zeroed by k_cull)
  uint32_t  pt_cap;
  float*    pt_terms;              // [gi_samples][gi_bounces * 3 + 1][N_SHARD * pt_cap]: bounce terms + their count per (sample, record)
  uint32_t  render_blocks;         // grid of k_render (number of class_hist rows in use)
  uint32_t  blocks_per_cu_cap;     // 0 = everything that is resident; frames that overlap others take 2 blocks per CU each
  unsigned long long* stats;       // [8]: rays_primary, rays_shadow, rays_bounce, box, tri, hitupd, pixels, drops
  unsigned long long* stats_partial;   // [rt_render_grid()*8] per-block partials of phase B
};

// Several frames of one shape in one frame pipeline (rt_render_batch_device): k_cull / k_compact run with the frame as
// grid.y over per-frame scratch regions, ONE persistent k_render walks all the frames' pixel lists, hottest class of every
// frame first.  RtBatch is passed by value next to RtRenderParams (whose U / outputs / cull rectangle are then unused); what
// differs between the frames lives in DEVICE memory, one 200-byte record per frame (a launch carries at most 4 KB of
// arguments: 16 records; rt_render_batch_device writes them with k_put_frames, 16 per launch, ahead of the pipeline).
constexpr int RT_U_FLOATS = 40;
struct RtFrameRec {
  float     U[RT_U_FLOATS];  // the uniform floats the kernels read: 0 .. 38 (compute.wgsl:4-13), padded to 40
  uint8_t*  rgba8;
  uint32_t* hit_tri;
  float*    hit_t;
  int       cull[4];         // x0, x1, y0, y1
  uint32_t  src;             // index of the first frame of the batch with the same 44 uniforms (its own index unless a duplicate view):
  uint32_t  pad_;            //   duplicates use that frame's tile masks and pixel list
};
constexpr int RT_FRAMES_PER_PUT = 16;
struct RtFrameChunk { RtFrameRec r[RT_FRAMES_PER_PUT]; };
struct RtBatch {
  uint32_t n;
  uint32_t mask_stride;      // u64 words between the frames' tile-mask arrays
  uint32_t bt_stride;        // u32 words between the frames' block totals
  uint32_t list_stride;      // u32 words between the frames' pixel lists
  uint32_t lc_stride;        // u32 words between the frames' list_count blocks
  const RtFrameRec* frames;  // [n], device memory
};
struct RtNoBatch {};
constexpr uint32_t RT_LC_WORDS = 32;   // list_count block of a frame: [0] entries, [1] hottest-class entries, [2..7] class thresholds,
                                       // [8 + c] start of class c in the list, [16 + c] entries of class c

// ---------------------------------------------------------------- build pipeline (rt_build.hip)
struct RtBuildBuffers {
  const float* pos; const float* nor; const uint32_t* vidx; const uint32_t* nidx;
  uint32_t n_tri;
  uint32_t* bounds_enc;     // [rt_bounds_words()] per-block ordered-uint partials: min.xyz / max.xyz
  float*    bounds_f;       // [8] `Bounds` uniform: min.xyz,0, invExtent.xyz,0 (morton_compute.wgsl:3-6)
  float*    centroids;      // [3T] (p0 + p1 + p2) / 3 per input triangle, k_bounds -> k_morton
  const float* tri_in_pos;  // [9T] the triangles' positions de-indexed, input order (k_pack_mesh, at upload): k_bounds streams them
  const float4* mesh_rec;   // [8T] the de-indexed mesh, one 128-byte record per input triangle (k_pack_mesh, at upload): positions, normals, texcoords, material
  uint32_t* codes;          // [T] unsorted Morton codes
  uint32_t* keys_a; uint32_t* keys_b;     // sort ping-pong
  uint32_t* vals_a; uint32_t* vals_b;
  uint32_t* hist;           // [rt_sort_hist_words()] global digit histograms (raw counts; scanned inside the sort passes)
  uint32_t* tile_status;    // [passes][n_tiles][digits] look-back words
  uint32_t* tile_counter;   // [rt_build_counter_words()] sort tickets, then the shard counters of cross_list
  uint32_t* err_flag;       // [1]
  int32_t*  span;           // [T-1] the other end j of node i's leaf range [min(i,j), max(i,j)]
  int32_t*  split;          // [T-1] split position of the nodes whose range crosses a 256-leaf block (the others never leave LDS)
  uint32_t* cross_list;     // [64 * rt_cross_capacity(T)] those nodes, as 64 compact lists
  float4*   refit_tables;   // [rt_refit_table_float4s(T)] block boxes, per-super-block sparse tables, super-block boxes
  uint32_t  n_sort_tiles;
  // optional surface inputs (NULL texcoords = none): the refit also writes the BVH-order records tri_uv [2T float4]
  const float* texcoords; const uint32_t* tex_idx; const int32_t* material_ids; float4* tri_uv;
};
uint32_t rt_bounds_words();
uint32_t rt_sort_tile_count(uint32_t n);
size_t   rt_sort_status_words(uint32_t n);
uint32_t rt_sort_hist_words();
void rt_launch_bounds(const RtBuildBuffers& b, hipStream_t s);
// experiment (RTCORE_WIDE_PAIRED): `wide` re-laid out as {node, likeliest child} pairs, one pair per 128-byte line
void rt_launch_pair_wide(const float4* wide, uint32_t n_int, uint32_t* up, uint32_t* pos, uint32_t* counter, float4* paired, hipStream_t s);
// mesh layout at upload: the indexed loadObj arrays de-indexed into one 128-byte record per triangle (tex_idx == NULL: no surface yet)
size_t rt_mesh_rec_float4s(uint32_t n_tri);
void rt_launch_pack_mesh(const float* pos, const float* nor, const uint32_t* vidx, const uint32_t* nidx, const float* texcoords,
                         const uint32_t* tex_idx, const int32_t* material_ids, uint32_t n_tri, float4* mesh_rec, float* tri_in_pos /* NULL: leave as is */, hipStream_t s);
void rt_launch_morton(const RtBuildBuffers& b, hipStream_t s);
void rt_launch_sort(const RtBuildBuffers& b, hipStream_t s);      // result in keys_a / vals_a
// also writes the traversal layout (wide, tri_pre) from the values it already holds: builds need no pack pass
void rt_launch_refit_local(const RtBuildBuffers& b, float* nodes, float* tri_pos, float* tri_nor, float4* wide, float4* tri_pre,
                           bool v2, hipStream_t s);
void rt_launch_refit_upper(const RtBuildBuffers& b, float* nodes, float4* wide, bool v2, hipStream_t s);
size_t rt_refit_table_float4s(uint32_t n_tri);
uint32_t rt_cross_capacity(uint32_t n_tri);
uint32_t rt_build_counter_words();
// derive the traversal layout from the interchange buffers; writes the root ref / error bits
void rt_launch_pack_impl(const RtBvhDev& bvh, uint32_t* d_root_ref, uint32_t* d_err, hipStream_t s);

// ---------------------------------------------------------------- traversal (rt_trace.hip)
#ifndef RT_N_CLASS
#define RT_N_CLASS 4              // cost classes of the pixel list (3 .. 8), see rt_trace.hip
#endif
uint32_t rt_render_grid(int n_cu);   // blocks of the persistent phase-B grid
uint32_t rt_cull_blocks(uint32_t n_tiles);
uint32_t rt_n_class();            // cost classes of the pixel list (rt_trace.hip N_CLASS)
uint32_t rt_shard_words();
uint32_t rt_pt_capacity(uint32_t n_pixels, uint32_t shard_group);   // records per PT sub-list that can never overflow
uint32_t rt_shard_stride();
// ev0/ev1 (optional) are recorded immediately before/after the traversal kernel proper
// want_stats == false skips the k_stats_final reduction (p.stats is then not written)
// p.tri_uv != NULL selects the textured instantiation
// ev_mid (optional): recorded behind the FIRST traversal launch (frames that run as traversal + shade launches)
void rt_launch_render(const RtRenderParams& p, int layout, bool count_work, bool want_stats, int n_cu, hipStream_t s,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, hipEvent_t ev_mid = nullptr);
// the same pipeline over bt.n frames (no stats, no PT split, compute.wgsl modes only)
// count_work: p.stats_partial = [bt.n][resident waves][8] u64 (zeroed here), p.stats = [bt.n][9] u64 totals (8 counters + pixels traced)
void rt_launch_render_batch(const RtRenderParams& p, const RtBatch& bt, int layout, bool count_work, int n_cu, hipStream_t s,
                            hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, hipEvent_t ev_mid = nullptr);
void rt_launch_put_frames(const RtFrameRec* host_recs, uint32_t n, RtFrameRec* d_frames, hipStream_t s);
// BVH-order surface records: record i <- input triangle order[i]
void rt_launch_gather_uv(const float* texcoords, const uint32_t* tex_idx, const int32_t* material_ids, const uint32_t* order,
                         uint32_t n_tri, float* tri_uv, hipStream_t s);
// Conservative screen-space bound of the root box (host, double precision): fills p.cull_*.
void rt_cull_rect(RtRenderParams& p, const float root_box[8]);
void rt_launch_assemble(const void* windows, void* image, uint32_t width, uint32_t height, int stripe_rows, int world, uint32_t col0,
                        uint32_t n_cols, uint32_t group0, uint32_t n_rows, size_t rank_stride, uint32_t fill, hipStream_t s);
void rt_launch_deinterleave(const void* packed, void* image, uint32_t width, uint32_t height, uint32_t elem_bytes,
                            int stripe_rows, int world, uint32_t rows_per_rank, hipStream_t s);
