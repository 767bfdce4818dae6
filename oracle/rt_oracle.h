/*
 * rt_oracle.h — scalar CPU ORACLE for the rtcore hot paths.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  librtcore.so (the product) never links,
 * loads or calls anything in oracle/.
 *
 * PARITY UNPINNED BY THE REFERENCE: 31415Hacker/RayTracer has no tests, golden vectors
 * or fixtures for this path, its WGSL cannot be compiled here, and
 * executing its script.js was denied.  The oracle is therefore a line-by-
 * line CPU restatement of the WGSL, pinned by hand-derived known answers
 * (tests/test_oracle_known_answers.py) and by structural property tests.
 *
 * Restated sources (all under /root/reference):
 *   web/morton_compute.wgsl:18-69   expandBits / quantize / main
 *   web/compute.wgsl:68-446         fetchCachedNode … cs_main
 *   web/script.js:207-227           normalizeMesh
 *   web/script.js:526-561           flattenForSSBO (node + triangle buffer layout)
 *   web/script.js:285-513           buildLBVH, the reference's CPU SAH builder (rt_oracle_sah.cpp)
 *   web/script.js:30-154,591-612,880-929   camera / TRS / invert4 / uniform packing
 *   web/shader.wgsl:132-440, shader.wgsl:136-454   the two older fragment-pipeline shaders
 * Sort / Karras radix tree / refit have no reference code; they follow
 * the definitions pinned there.
 *
 * Build: g++ -O2 -ffp-contract=off (no fast-math), see oracle/Makefile.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODE_PRIMARY = 0, ORC_MODE_SHADOW = 1, ORC_MODE_PT = 2,
       ORC_MODE_FRAG_WEB = 3,     /* web/shader.wgsl  computeLighting_optimized (:369-433) under fs_main (:435-440) */
       ORC_MODE_FRAG_ROOT = 4 };  /* shader.wgsl      computeLighting_optimized (:387-447) under fs_main (:449-454) */

typedef struct orc_render_opts {
  int32_t mode;          /* ORC_MODE_* */
  int32_t gi_samples;    /* compute.wgsl:26 GI_SAMPLES (32); also drives the shadow stratification grid */
  int32_t gi_bounces;    /* compute.wgsl:27 GI_BOUNCES (3) */
  int32_t soft_samples;  /* compute.wgsl:25 SOFT_SAMPLES (4) */
  int32_t max_stack;     /* compute.wgsl:259 MAX_STACK (12), <= 64 */
  int32_t row_begin;     /* rows [row_begin,row_end) step row_step are rendered; others untouched */
  int32_t row_end;
  int32_t row_step;
  int32_t n_threads;     /* std::thread row striping; 1 = scalar */
  int32_t node_encoding; /* 0/1 = v1 records (indices as f32 values, reference), 2 = v2 (u32 bit-casts) */
} orc_render_opts;

typedef struct orc_render_stats {
  uint64_t rays_primary, rays_shadow, rays_bounce;
  uint64_t box_tests;    /* V: (ray,node) pairs whose box is tested, each counted once */
  uint64_t tri_tests;    /* K: triangles fetched in leaves */
  uint64_t hit_updates;  /* H: closer-hit updates (normal fetches) */
  uint64_t pixels;       /* P: pixels written */
  uint64_t stack_drops;  /* far children dropped because the stack was full (F9) */
  double   seconds;
} orc_render_stats;

/* web/morton_compute.wgsl:18-31 */
uint32_t orc_expand_bits(uint32_t v);
uint32_t orc_quantize(float x);
/* web/compute.wgsl:135-149 */
uint32_t orc_hash_u32(uint32_t seed);
float    orc_rand2(float sx, float sy);
/* implementation-pinned transcendental kernels (DESIGN.md "numerics contract") */
void  orc_sincos(float theta, float* s, float* c);
float orc_pow(float x, float y);
float orc_acos(float x);
/* RNGs of the older shaders: web/shader.wgsl:176-185 (variant 1), shader.wgsl:320-331 (variant 2) */
uint32_t orc_frag_hash(int variant, uint32_t x);
float    orc_frag_rand2(int variant, float sx, float sy);

/* web/script.js:207-227 — in place */
void orc_normalize_mesh(float* pos, uint32_t n_vert);

/* min / 1/(max-min) of triangle centroids (0 where extent is 0) */
void orc_centroid_bounds(const float* pos, const uint32_t* vidx, uint32_t n_tri,
                         float bmin[3], float inv_extent[3]);
/* web/morton_compute.wgsl:33-69 */
void orc_morton(const float* pos, const uint32_t* vidx, uint32_t n_tri,
                const float bmin[3], const float inv_extent[3], uint32_t* codes);
/* stable sort of (code, index) */
void orc_sort(const uint32_t* codes, uint32_t n_tri, uint32_t* sorted_codes, uint32_t* order);
/* Karras radix tree over sorted codes. Node ids: internal i -> i, leaf j -> (T-1)+j.
 * child0/child1 have T-1 entries; parent has 2T-1 entries (parent[0] = -1). */
void orc_karras(const uint32_t* sorted_codes, uint32_t n_tri,
                int32_t* child0, int32_t* child1, int32_t* parent);
/* Whole build: bounds -> morton -> sort -> tree -> refit -> flat buffers.
 * nodes: 8 floats per node, (2T-1) nodes, layout of web/script.js:546-559.
 * tri_pos/tri_nor: 12 floats per triangle (3 x vec4, w = 0), BVH order (script.js:527-545).
 * codes (unsorted) and order may be NULL. */
void orc_build_lbvh(const float* pos, uint32_t n_vert, const float* nor, uint32_t n_nor,
                    const uint32_t* vidx, const uint32_t* nidx, uint32_t n_tri,
                    float* nodes, float* tri_pos, float* tri_nor,
                    uint32_t* codes, uint32_t* order, double* seconds);

/* Same with the node-record encoding chosen: 1 = v1 (reference: indices as f32 values, exact below
 * 2^24 nodes), 2 = v2 (u32 bit-casts in the same slots, leaf flag = bit 31 of B, low bits = count). */
void orc_build_lbvh_enc(const float* pos, uint32_t n_vert, const float* nor, uint32_t n_nor,
                        const uint32_t* vidx, const uint32_t* nidx, uint32_t n_tri,
                        float* nodes, float* tri_pos, float* tri_nor,
                        uint32_t* codes, uint32_t* order, double* seconds, int encoding);

/* The reference's ACTUAL builder: web/script.js:285-513 `buildLBVH` — a recursive 20-bin SAH builder
 * over centroids with median fallback, DFS pre-order numbering — followed by flattenForSSBO (script.js:526-561).
 * JS double arithmetic with Float32Array rounding points reproduced.  nodes: capacity nodes_capacity records of 8
 * floats (v1 encoding; at leaf = 1 the tree has 2T-1 nodes).  Returns 0, -2 if capacity is too small (*n_nodes_out
 * is still set), -11 where the reference would recurse forever (partition left one side empty), -12 on NaN input. */
int orc_build_sah(const float* positions, const float* normals, const uint32_t* vertIdx, const uint32_t* normIdx,
                  uint32_t n_tri, int leaf, float* nodes_out, uint32_t nodes_capacity, uint32_t* n_nodes_out,
                  float* tri_pos, float* tri_nor, uint32_t* order_out, double* seconds);

/* web/script.js:30-154,591-612,880-929: the 44 uniform floats.
 * sliders = {scaleX,scaleY,scaleZ, rotX,rotY,rotZ (degrees), posX,posY,posZ} */
void orc_make_uniforms(const double cam_pos[3], double yaw, double pitch,
                       const double sliders[9], uint32_t width, uint32_t height,
                       float time_value, float out44[44]);

/* web/compute.wgsl:413-446 cs_main over the selected rows.
 * rgba8: 4*w*h bytes; hit_tri: w*h (0xFFFFFFFF = miss, else BVH-order triangle index of the
 * primary hit); hit_t: w*h (1e20 on miss).  Any output may be NULL. */
int orc_render(const float* nodes, uint32_t n_nodes, const float* tri_pos, const float* tri_nor,
               uint32_t n_tri, const float uniforms[44], const orc_render_opts* opts,
               uint8_t* rgba8, uint32_t* hit_tri, float* hit_t, orc_render_stats* stats);

/* Material / texture extension (no reference counterpart: compute.wgsl:372 is a constant albedo; the
 * sampling rule is written out at rt_oracle.cpp:hit_albedo and include/rtcore.h).  While a surface is set, orc_render
 * takes the albedo of every hit from it; NULL restores the reference's constant. */
typedef struct orc_surface {
  const float*    tri_uv;       /* [8T], BVH order: u0 v0 u1 v1 u2 v2 bits(material id, i32) 0 */
  const float*    materials;    /* [4M]: Kd.r Kd.g Kd.b bits(texture slot, 0xFFFFFFFF = none) */
  uint32_t        n_materials;
  const uint32_t* texels;       /* all textures back to back, one u32 per texel: R | G<<8 | B<<16 | 255<<24, row 0 = top */
  const uint32_t* tex_table;    /* [4 per texture]: first texel, width, height, 0 */
  uint32_t        n_textures;
} orc_surface;
void orc_set_surface(const orc_surface* s);
void orc_gather_tri_uv(const float* texcoords, const uint32_t* tex_idx, const int32_t* material_ids /* or NULL */,
                       const uint32_t* order, uint32_t n_tri, float* tri_uv);

/* analysis aid: when non-NULL, later orc_render calls also write box tests per pixel (w*h) */
void orc_set_work_image(uint32_t* img);
/* analysis aid (single-threaded renders only): counts closest-hit internal steps per node */
void orc_set_visit_counts(uint32_t* per_node);

#ifdef __cplusplus
}
#endif
#endif
