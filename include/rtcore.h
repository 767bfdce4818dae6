/*
 * rtcore.h — C ABI of the MI355X-native ray-tracing core (librtcore.so).
 *
 * Drop-in boundary for the two hot paths of 31415Hacker/RayTracer.  The
 * reference has no FFI; the seam is its data contract, crossed twice:
 *   build side : loadObj() mesh arrays            web/script.js:198-203
 *                -> buildLBVH(mesh, 1)            web/script.js:285-513 (call site :729)
 *                -> flattenForSSBO()              web/script.js:526-561
 *   render side: bind group 0 + 176-byte uniforms web/compute.wgsl:4-22, web/script.js:798-810,899-930
 *                -> one dispatch of cs_main       web/compute.wgsl:413-446, web/script.js:936-945
 * Every entry point names the reference interface it replaces.
 *
 * Conventions: plain C types only; 0 = success, negative = error (message through
 * rt_last_error, valid until the next call on the same context); inputs are copied
 * during the call; outputs are caller-allocated; a context is not thread-safe (calls are
 * serialised by the caller, like the reference's single JS thread); one context drives
 * one GPU (one process per GPU; multi-GPU sharding is done by the host with RCCL, see
 * the *_device entry points and INTEGRATION.md).
 *
 * There is NO CPU fallback behind this ABI: every compute entry point fails with
 * RT_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef RTCORE_H
#define RTCORE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5   /* 5: rt_material carries all of material_t + texture options, rt_batch_stats, rt_profile_end_phases, rt_debug_device_allocations;
                              2: rt_render_stats.pixels_traced, rt_render_device_noalloc, cache-file trailer; 3: rt_render_batch_device; 4: rt_frame_rect, rt_assemble_device_on, RT_MAX_BATCH 32 */

enum {
  RT_OK = 0,
  RT_ERR_INVALID = -1,      /* bad argument / call order */
  RT_ERR_NO_DEVICE = -2,    /* HIP runtime or device unavailable */
  RT_ERR_HIP = -3,          /* a HIP call failed (see rt_last_error) */
  RT_ERR_IO = -4,           /* file / parse error */
  RT_ERR_LIMIT = -5,        /* scene exceeds an encoding limit (F11: v1 node indices < 2^24) */
  RT_ERR_INTERNAL = -6      /* a kernel reported an internal failure (e.g. bounded spin expired) */
};

/* ------------------------------------------------------------------ context */
typedef struct rt_ctx rt_ctx;

/* Replaces: adapter/device acquisition, web/script.js:706-715.
 * backend must be "hip" (NULL = "hip"); device_ids/n_devices: exactly one device id
 * (NULL/0 = device 0). */
int         rt_create(rt_ctx** out, const char* backend, const int* device_ids, int n_devices);
void        rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);
int         rt_abi_version(void);
/* Diagnostic: how many hipMalloc / hipFree calls this library has made in the process so far (all contexts).  Unchanged
 * across a call = that call allocated and freed nothing; tests/test_zz_no_alloc.py holds rt_render_device_noalloc to it. */
uint64_t    rt_debug_device_allocations(void);

/* ------------------------------------------------------------------ scene input (host side, no GPU needed) */
/* OBJ/MTL ingest with tiny_obj_loader v2.0.0 semantics (reference: /tiny_obj_loader.h:638,
 * float parser :954-1122, index fix-up :867-907, quad split along the shorter diagonal
 * :1701-1813); replaces loadObj, web/script.js:164-204.  Output arrays follow the loadObj
 * return contract (script.js:198-203) plus texcoords / per-face material ids. */
typedef struct rt_mesh rt_mesh;
typedef struct rt_mesh_view {
  const float*    positions;  uint32_t n_vert;   /* 3 floats each */
  const float*    normals;    uint32_t n_nor;    /* 3 floats each (one dummy (0,0,0) entry if the file has none) */
  const float*    texcoords;  uint32_t n_tex;    /* 2 floats each (one dummy (0,0) entry if the file has none) */
  const uint32_t* vert_idx;                      /* 3 per triangle */
  const uint32_t* nor_idx;                       /* 3 per triangle */
  const uint32_t* tex_idx;                       /* 3 per triangle */
  const int32_t*  material_ids;                  /* 1 per triangle, -1 = none */
  uint32_t        n_tri;
  uint32_t        n_materials;
} rt_mesh_view;
/* tinyobj::texture_option_t (tiny_obj_loader.h:166-186), filled by ParseTextureNameAndOption (:1411-1528): the options of a
 * map_* / bump / disp / refl / norm line that stand IN FRONT of the file name (what follows the name belongs to the name). */
typedef struct rt_texopt {
  int32_t type;               /* -type: 0 none, 1 sphere, 2 cube_top, 3 cube_bottom, 4 cube_front, 5 cube_back, 6 cube_left, 7 cube_right */
  float   sharpness;          /* -boost (1) */
  float   brightness;         /* -mm base (0) */
  float   contrast;           /* -mm gain (1) */
  float   origin_offset[3];   /* -o u v w (0 0 0) */
  float   scale[3];           /* -s u v w (1 1 1) */
  float   turbulence[3];      /* -t u v w (0 0 0) */
  int32_t texture_resolution; /* -texres (-1) */
  int32_t clamp;              /* -clamp on|off (0) */
  int32_t blendu, blendv;     /* -blendu / -blendv on|off (1) */
  float   bump_multiplier;    /* -bm (1) */
  char    imfchan;            /* -imfchan: 'l' for the bump map, 'm' otherwise */
  char    colorspace[31];     /* -colorspace (empty) */
} rt_texopt;
/* the thirteen texture names of material_t, in the order LoadMtl tests their keys (:2600-2747) */
enum { RT_MAP_KA = 0 /* map_Ka */, RT_MAP_KD /* map_Kd */, RT_MAP_KS /* map_Ks */, RT_MAP_NS /* map_Ns */,
       RT_MAP_BUMP /* map_bump, map_Bump, bump */, RT_MAP_DISP /* map_disp, map_Disp, disp */, RT_MAP_D /* map_d */, RT_MAP_REFL /* refl */,
       RT_MAP_PR /* map_Pr */, RT_MAP_PM /* map_Pm */, RT_MAP_PS /* map_Ps */, RT_MAP_KE /* map_Ke */, RT_MAP_NORM /* norm */, RT_N_MAPS };
typedef struct rt_material {                     /* tinyobj::material_t (tiny_obj_loader.h:188-247), as LoadMtl (:2322-2773) fills it */
  char  name[128];                               /* newmtl: ONE word (parseString) */
  float diffuse[3];                              /* Kd; (0,0,0) without a Kd line, (0.6,0.6,0.6) when a map_Kd line comes before any Kd (:2617) */
  float emission[3];                             /* Ke */
  float dissolve;                                /* d, or 1 - Tr when no d came before it (`d` wins, :2524) */
  int   illum;
  char  diffuse_texname[256];                    /* = texname[RT_MAP_KD] */
  char  bump_texname[256];                       /* = texname[RT_MAP_BUMP] */
  /* ABI 5: the rest of material_t.  Names longer than 255 bytes are cut (unbounded strings there). */
  float ambient[3], specular[3], transmittance[3];   /* Ka, Ks, Kt / Tf */
  float shininess, ior;                          /* Ns (1), Ni (1) */
  float roughness, metallic, sheen, clearcoat_thickness, clearcoat_roughness, anisotropy, anisotropy_rotation;   /* Pr Pm Ps Pc Pcr aniso anisor */
  char  texname[RT_N_MAPS][256];
  rt_texopt texopt[RT_N_MAPS];
} rt_material;

int  rt_mesh_load_obj(const char* obj_path, const char* mtl_search_dir /* NULL = dir of obj */,
                      rt_mesh** out, char* err, size_t err_len);
int  rt_mesh_parse_obj(const char* obj_text, size_t obj_len, const char* mtl_text /* or NULL */, size_t mtl_len,
                       rt_mesh** out, char* err, size_t err_len);
int  rt_mesh_view_get(const rt_mesh* m, rt_mesh_view* out);
int  rt_mesh_material(const rt_mesh* m, uint32_t i, rt_material* out);
/* material_t::unknown_parameter (tiny_obj_loader.h:247, filled at :2749-2762): the `key value` lines of the MTL that
 * tinyobj has no field for, in the order of its ordered map (sorted by key; the first value of a repeated key wins).  The strings
 * live as long as the mesh. */
uint32_t rt_mesh_material_n_unknown(const rt_mesh* m, uint32_t i);
int  rt_mesh_material_unknown(const rt_mesh* m, uint32_t i, uint32_t k, const char** key, const char** value);
void rt_mesh_free(rt_mesh* m);

/* Replaces normalizeMesh, web/script.js:207-227 (in place; double math, f32 store). */
int  rt_normalize_mesh(float* positions, uint32_t n_vert);

/* Replaces updateCamera + basis + TRS sliders + invert4 + udata packing,
 * web/script.js:591-612, 880-896, 136-154, 30-133, 899-929.  out44 = the 176-byte block
 * of web/compute.wgsl:4-13 (cameraPos@0 lightPos@4 camRight@8 camUp@12 camForward@16
 * model(inverse TRS)@20 resolution@36 time@38, float indices). */
typedef struct rt_camera {
  double pos[3];         /* script.js:4  default (0,0,5) */
  double yaw, pitch;     /* script.js:5-6 */
  double scale[3];       /* sliders scaleX/Y/Z, default 1 */
  double rot_deg[3];     /* sliders rotX/Y/Z in degrees, default 0 (rotX gets -pi, script.js:148) */
  double translate[3];   /* sliders posX/Y/Z, default 0 */
  uint32_t width, height;
  float  time;           /* the reference always uploads 0 */
} rt_camera;
void rt_camera_default(rt_camera* cam, uint32_t width, uint32_t height);
int  rt_make_uniforms(const rt_camera* cam, float out44[44]);

/* ------------------------------------------------------------------ hot path 1: LBVH build */
/* Replaces the makeBuf uploads of the raw mesh (script.js:571-578): copies the loadObj
 * arrays to the device. */
int rt_set_mesh(rt_ctx* ctx, const float* positions, uint32_t n_vert, const float* normals, uint32_t n_nor,
                const uint32_t* vert_idx, const uint32_t* nor_idx, uint32_t n_tri);

/* Node-record encoding of slots A / B of the 32-byte record (flattenForSSBO, script.js:546-559).
 * RT_NODES_V1 (default) is the reference's: child / triangle indices as f32 VALUES, leaf B = -count; exact
 * only below 2^24 nodes, larger builds return RT_ERR_LIMIT.  RT_NODES_V2 keeps
 * the record and its size and stores u32 bit-casts in the same slots: internal A = child0, B = child1
 * (bit 31 clear); leaf A = tri_start, B = 0x80000000 | tri_count.  Up to 2^28 triangles.  The setting applies
 * to the next rt_build_lbvh / rt_set_bvh / rt_load_bvh / rt_bvh_adopt_device and to what rt_get_nodes,
 * rt_save_bvh and rt_bvh_export_device hand out (the buffer is kept in the encoding it was made in). */
enum { RT_NODES_V1 = 1, RT_NODES_V2 = 2 };
int rt_set_node_encoding(rt_ctx* ctx, int encoding);
int rt_get_node_encoding(rt_ctx* ctx);   /* of the current BVH if there is one, else the setting */

typedef struct rt_build_stats {
  /* ms_total = hipEvent time from the build's first launch to its last (+ ms_pack).  The stage fields are filled only after
   * rt_set_build_stage_timing(ctx, 1) (0 otherwise): the events between the launches cost the build ~2 us each, so a build
   * that reports its stages is ~3 % slower at 1.3 M triangles and its ms_total is the sum of the stages.
   * hipEvent times of the build's stages: ms_bounds = centroid bounds; ms_morton = codes + the sort's digit histograms;
   * ms_sort = the three 10-bit onesweep passes; ms_refit = leaf records, triangle buffers and every node whose leaves
   * lie inside one 256-leaf block (Karras search + boxes, k_refit_local); ms_tree = the nodes that span blocks (their
   * Karras search and boxes, k_refit_mid + k_refit_spans); ms_pack = 0 for builds (the traversal records are written by
   * the refit kernels; only imported trees are packed). */
  float ms_bounds, ms_morton, ms_sort, ms_tree, ms_refit, ms_pack, ms_total;
  uint32_t n_tri, n_nodes;
  uint64_t algorithmic_bytes;   /* 560 B x n_tri */
  uint32_t refit_fallbacks;     /* always 0 since round 2: the refit has no hand-off between workgroups left that could fail (kept for ABI stability) */
} rt_build_stats;
/* Per-stage build times on / off (default off; RTCORE_BUILD_STAGE_TIMES=0/1 overrides).  No counterpart in the reference. */
int rt_set_build_stage_timing(rt_ctx* ctx, int on);

/* Replaces buildLBVH(raw, 1) + flattenForSSBO (script.js:729-730) with the Morton LBVH the
 * reference only sketched (web/morton_compute.wgsl, compiled but never dispatched — F4):
 * centroid bounds -> 30-bit Morton -> stable onesweep radix sort -> Karras radix tree ->
 * atomic bottom-up refit -> flat node / triangle buffers. */
int rt_build_lbvh(rt_ctx* ctx, rt_build_stats* out /* may be NULL */);

/* Introspection for parity tests (device -> caller-allocated host arrays). */
int rt_get_bounds(rt_ctx* ctx, float bmin[3], float inv_extent[3]);   /* `Bounds` uniform, morton_compute.wgsl:3-6 */
int rt_get_morton(rt_ctx* ctx, uint32_t* codes /* [T], triangle order of the input */);
int rt_get_tri_order(rt_ctx* ctx, uint32_t* order /* [T]: BVH-order position -> input triangle */);
int rt_get_counts(rt_ctx* ctx, uint32_t* n_tri, uint32_t* n_nodes);
int rt_get_nodes(rt_ctx* ctx, float* nodes /* [8*N], layout script.js:546-559 */);
int rt_get_triangles(rt_ctx* ctx, float* tri_pos /* [12*T] */, float* tri_nor /* [12*T] */); /* script.js:527-545 */

/* Replaces the bvh/tri makeBuf uploads (script.js:743-748): import an externally built flat
 * BVH (e.g. the reference's SAH builder output, or a BVH cache file) in the v1 layout. */
int rt_set_bvh(rt_ctx* ctx, const float* nodes, uint32_t n_nodes,
               const float* tri_pos, const float* tri_nor, uint32_t n_tri);

/* BVH cache file (what web/bvh-buffer.bin was meant to be): little-endian f32
 * nodes[8N] then tri_pos[12T] then tri_nor[12T].  A v1 tree with N = 2T-1 (every LBVH build, the reference's builder at
 * leaf size 1) is written header-less, so the file is the raw dump the reference's bindings 4/5 take.  Any other tree
 * (multi-triangle leaves as flattenForSSBO writes them, script.js:546-559; v2 records) gets a 16-byte trailer
 * {u32 magic "RTBV", n_nodes, n_tri, encoding}; rt_load_bvh reads both forms, and a trailer's encoding overrides (and
 * becomes) the context's rt_set_node_encoding setting. */
int rt_save_bvh(rt_ctx* ctx, const char* path);
int rt_load_bvh(rt_ctx* ctx, const char* path);

/* ------------------------------------------------------------------ hot path 2: traversal + shading */
enum { RT_MODE_PRIMARY = 0,   /* primary ray, N.L shading, no shadow rays (BASELINE cfg 1-2) */
       RT_MODE_SHADOW  = 1,   /* primary + SOFT_SAMPLES shadow rays (cfg 3) */
       RT_MODE_PT      = 2,   /* full computeLighting_optimized (cfg 4) */
       /* The two older fragment-pipeline shaders, same node / triangle buffers and traversal loop.  They
        * differ from compute.wgsl in the triangle test's thresholds, plain mul/add where compute.wgsl writes fma, a sky
        * pixel (0.6, 0.8, 1, 1) on a miss, an 8-tap Poisson-disk area light (radius = lightPos.w, uniforms[7]), their RNGs
        * and GI estimators, and the image's y direction: row 0 is the TOP framebuffer row, uv.y = 1 - (y + 0.5) / height
        * (a rasteriser's interpolation of uv is implementation-defined in its last bits; that expression is the definition
        * here).  Only uniforms[0..35] are read besides the resolution.  Constant albedo, RT_LAYOUT_WIDE64 only. */
       RT_MODE_FRAG_WEB  = 3, /* web/shader.wgsl:369-440 : gi_samples x gi_bounces (32 x 3) cosine-weighted GI, bounce rays end at t >= 50 */
       RT_MODE_FRAG_ROOT = 4 };/* shader.wgsl:387-454 : <= 16 fixed stratified directions (gi_bounces ignored), one bounce, roulette 0.9 */
enum { RT_LAYOUT_WIDE64 = 0,  /* default: 64-byte two-child records + precomputed triangle edges */
       RT_LAYOUT_AOS32  = 1 };/* traverse the 32-byte interchange records directly (A/B baseline) */
enum { RT_FLAG_COUNT_WORK = 1,   /* also count box/triangle tests (slower; for parity + roofline bytes) */
       RT_FLAG_TEXTURED   = 2 }; /* albedo of every hit from the bound surface (materials + map_Kd textures) instead of
                                    the constant of compute.wgsl:372; see "material / texture shading" below */

typedef struct rt_render_opts {
  int32_t mode;           /* RT_MODE_* */
  int32_t gi_samples;     /* compute.wgsl:26 (32) */
  int32_t gi_bounces;     /* compute.wgsl:27 (3)  */
  int32_t soft_samples;   /* compute.wgsl:25 (4)  */
  int32_t max_stack;      /* compute.wgsl:259 (12); 2..32 */
  int32_t layout;         /* RT_LAYOUT_* */
  int32_t flags;          /* RT_FLAG_* */
  int32_t stripe_rows;    /* image-space sharding: rows per stripe (16); 0 = whole image */
  int32_t stripe_rank;    /* this context renders stripes s with s % stripe_world == stripe_rank */
  int32_t stripe_world;
} rt_render_opts;
void rt_render_opts_default(rt_render_opts* o);
/* the constants the shader of `mode` is written with (compute.wgsl:25-27,259; web/shader.wgsl:22-24,273; shader.wgsl:24-25,191) */
int rt_render_opts_for_mode(rt_render_opts* o, int mode);

typedef struct rt_render_stats {
  uint64_t rays_primary, rays_shadow, rays_bounce;
  uint64_t box_tests, tri_tests, hit_updates;   /* only with RT_FLAG_COUNT_WORK */
  uint64_t pixels;
  uint64_t stack_drops;                          /* only with RT_FLAG_COUNT_WORK */
  float    ms_kernel;                            /* hipEvent time of the frame pipeline (cull + compact + traversal + stats) */
  float    ms_traverse;                          /* hipEvent time of the traversal kernel (k_render) alone */
  uint64_t pixels_traced;                        /* pixels whose primary ray enters the root box: the ones the traversal kernel
                                                    traces.  The other pixels - pixels_traced are final after the cull pass and are
                                                    counted as one primary ray + one root-box test + one pixel each, so that the
                                                    counters equal the reference's per-pixel loop (compute.wgsl:268-275) */
} rt_render_stats;

/* Replaces queue.writeBuffer(uniformBuf) + dispatchWorkgroups(ceil(w/16), ceil(h/16)) of
 * cs_main (script.js:930-945).  rgba8 = the rgba8unorm storage texture (compute.wgsl:22,445),
 * row 0 first.  hit_tri (BVH-order index of the primary hit, 0xFFFFFFFF = miss) and hit_t
 * are extra outputs for parity (the reference returns neither).  Outputs may be NULL.
 * With striping, only this rank's rows of the full-size host images are written. */
int rt_render(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts,
              uint8_t* rgba8 /* [4*w*h] */, uint32_t* hit_tri /* [w*h] */, float* hit_t /* [w*h] */,
              rt_render_stats* out /* may be NULL */);

/* Same, outputs in caller-provided DEVICE memory of this context's GPU, asynchronous on
 * the context stream unless stats != NULL.  With striping the outputs are PACKED: local
 * row = (stripe / stripe_world) * stripe_rows + row_in_stripe (rt_packed_rows gives the
 * row count).  The three pointers must be 4-byte aligned (pixels are stored as 32-bit words; RT_ERR_INVALID otherwise);
 * 16-byte alignment lets whole culled rows go out as 16-byte stores. */
int rt_render_device(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts,
                     void* d_rgba8, void* d_hit_tri, void* d_hit_t, rt_render_stats* out);
uint32_t rt_packed_rows(uint32_t height, int32_t stripe_rows, int32_t stripe_rank, int32_t stripe_world);

/* rt_render_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference (a WebGPU page never sees an allocator).  Same
 * kernels, same results as rt_render_device bit for bit; the frame's scratch (compacted pixel list, tile masks, queue
 * heads, per-block counters) lives in ONE static device array of 38.6 MB that every GPU the library runs on holds until
 * the process ends, instead of the lazily grown allocations of rt_render_device.  Calls no hipMalloc / hipFree.
 * Limits: local_rows x width <= 3840 x 2160 pixels (the whole image, or this rank's stripe share) and at most 145 984
 * 8x8 tiles - RT_ERR_LIMIT beyond that; asynchronous on rt_stream(), always (frames in flight do not apply, there are no
 * stats); no frame-coherent scheduling hints (frames run at the unhinted speed); RT_MODE_PT runs one lane per pixel (the
 * PT split needs large term buffers); with RT_FLAG_TEXTURED the material and texture tables must already exist
 * (rt_set_materials / rt_set_texture).  The array is shared by every context on the same GPU and keeps the last call's
 * contents (each call rewrites what it reads): calls that overlap on one GPU - two contexts, two processes' worth of
 * streams - race; serialise them. */
int rt_render_device_noalloc(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts,
                             void* d_rgba8, void* d_hit_tri, void* d_hit_t);
/* n_frames (1 .. RT_MAX_BATCH) frames of ONE image shape in one pipeline: what the reference does as n_frames successive
 * frame() calls (script.js:865-945: writeBuffer(uniformBuf) + dispatchWorkgroups per animation frame) - e.g. the next
 * frames of a camera path, or the same frame's successive PT time seeds.  uniforms = n_frames x 44 floats; every frame
 * must carry the same resolution (RT_ERR_INVALID otherwise); opts (mode, sample counts, striping, layout, flags) apply to
 * all of them.  d_rgba8 / d_hit_tri / d_hit_t are HOST arrays of n_frames device pointers (an array, or single entries,
 * may be NULL); frame f's outputs are exactly those of rt_render_device(uniforms + 44 f, ...), bit for bit.  The frames'
 * cull and compaction passes run as one launch each and ONE persistent traversal launch drains all the frames' pixel
 * lists (the hottest pixels of every frame first), so the launch overheads and the idle tail of a frame are paid once
 * per batch: this is the entry point for small frames / stripe shares (a 1/8 share of a 4K frame is ~0.1 ms of work).
 * Per frame of a 32-frame batch against per-frame launches (one MI355X): 4K 0.44 -> 0.26 ms, 1080p 0.26 -> 0.064 ms.
 * (What differs between the frames - uniforms, output pointers - is copied into device memory ahead of the pipeline: the
 * uniforms and pointer arrays are free again when the call returns.)
 * Every mode (PT runs one lane per pixel; RT_MODE_FRAG_WEB / FRAG_ROOT as in rt_render_device: WIDE64, untextured); with RT_FLAG_COUNT_WORK every frame's counters are kept
 * (rt_batch_stats below; the counting instantiation is a few per cent slower, as for single frames); frames whose 44 uniforms
 * are byte-identical share one cull / compaction result (each is still traced and written in full);
 * asynchronous on rt_stream() - or, with rt_set_frames_in_flight(n >= 2), on the frame streams under the ordering rules
 * given there (a batch counts as one frame: the tail of batch i overlaps the head of batch i+1); the scheduling hints
 * left behind are those of the last frame, and a batch whose views differ is not ordered by hints at all. */
#define RT_MAX_BATCH 32
int rt_render_batch_device(rt_ctx* ctx, uint32_t n_frames, const float* uniforms /* [n_frames][44] */, const rt_render_opts* opts,
                           void* const* d_rgba8, void* const* d_hit_tri, void* const* d_hit_t);
/* The counters of the last rt_render_batch_device call that carried RT_FLAG_COUNT_WORK: one rt_render_stats per frame, each
 * equal to what rt_render_device(uniforms + 44 f, ...) reports for that frame (ms_kernel / ms_traverse are 0: a batch is one
 * launch; rt_profile_begin/end time it).  n_frames must be that batch's frame count.  Synchronises. */
int rt_batch_stats(rt_ctx* ctx, uint32_t n_frames, rt_render_stats* out /* [n_frames] */);
/* De-interleave `world` packed stripe buffers (concatenated rank-major, each
 * rows_per_rank x width x elem_bytes) into a full image; all pointers are device memory. */
int rt_deinterleave_device(rt_ctx* ctx, const void* d_packed, void* d_image, uint32_t width, uint32_t height,
                           uint32_t elem_bytes, int32_t stripe_rows, int32_t stripe_world, uint32_t rows_per_rank);
/* Same on a caller's hipStream_t of this GPU (NULL = rt_stream()): lets the exchange leg of frame i (collective +
 * de-interleave) run beside the traversal of frame i+1, which owns the context stream. */
int rt_deinterleave_device_on(rt_ctx* ctx, void* stream, const void* d_packed, void* d_image, uint32_t width, uint32_t height,
                              uint32_t elem_bytes, int32_t stripe_rows, int32_t stripe_world, uint32_t rows_per_rank);
/* Exchange of WINDOWS instead of whole stripe buffers (no counterpart in the reference: one GPU draws the whole canvas,
 * web/script.js:936-945).  Most of a frame is background: pixels outside the screen rectangle of the tree's root box
 * are misses whatever the scene holds (the cull pass of rt_render* writes them without tracing a ray).  rt_frame_rect
 * returns that rectangle for the compute.wgsl modes ({x0, y0, x1, y1}, inclusive; x1 < x0 or y1 < y0: nothing of the tree is on screen) and the value
 * of a missed pixel, so every rank can compute the same window from the uniforms alone and send only
 *   columns [col0, col0 + n_cols) of its packed rows [group0 * stripe_rows, group0 * stripe_rows + n_rows)
 * (stripe group g = the image rows [g * world * stripe_rows, (g + 1) * world * stripe_rows): one stripe of every rank).
 * rt_assemble_device_on builds the full image from the gathered windows (rank r's window, n_rows x n_cols elements of 4
 * bytes, at d_windows + r * rank_stride elements) and `fill` everywhere else.  cfg 3's camera: 9 MB instead of 33 MB per
 * frame cross the links.  raytracer_amd/dist.py BatchedFrames exchanges the union window of a batch's frames. */
int rt_frame_rect(rt_ctx* ctx, const float uniforms[44], int32_t rect[4], uint32_t* miss_rgba);
int rt_assemble_device_on(rt_ctx* ctx, void* stream /* NULL = rt_stream() */, const void* d_windows, void* d_image, uint32_t width,
                          uint32_t height, int32_t stripe_rows, int32_t stripe_world, uint32_t col0, uint32_t n_cols, uint32_t group0,
                          uint32_t n_rows, uint64_t rank_stride, uint32_t fill);

/* ------------------------------------------------------------------ material / texture shading (extension) */
/* The reference shades every surface with one constant albedo (compute.wgsl:372; no WGSL reads a vt, texture or
 * material) although its scenes carry MTL files and textures (models/ *.mtl, map_Kd).  north_star asks for "the
 * simple material/texture shading"; with RT_FLAG_TEXTURED the albedo at a hit is defined as
 *   material id < 0 or >= n_materials            -> the reference constant (1, 0.84, 0)
 *   albedo = Kd, times the map_Kd texel when the material names a texture
 *   uv     = fma(uv2, v, fma(uv1, u, uv0 * ((1 - u) - v)))     (u, v: Moeller-Trumbore barycentrics; the form of the
 *                                                               normal interpolation, compute.wgsl:291)
 *   f = uv - floor(uv) (repeat);  x = min(u32(f.x * W), W-1);  y = min(u32((1 - f.y) * H), H-1)   (OBJ v = 0 = bottom row)
 *   nearest texel, channel = f32(byte) / 255, no filtering, no colour-space conversion
 * and it replaces `albedo` at the primary hit (compute.wgsl:372,378,408) and, per bounce, the factor of
 * `throughput *= albedo` (:400) by the albedo of the surface the bounce ray hit.  Everything else is unchanged. */
typedef struct rt_material_rec {
  float   kd[3];       /* tinyobj material_t::diffuse (Kd) */
  int32_t texture;     /* texture slot of map_Kd, -1 = none */
} rt_material_rec;
#define RT_MAX_TEXTURES 64

/* Per-corner texcoords and per-face material ids of the mesh given to rt_set_mesh (the vt / usemtl data loadObj drops,
 * script.js:176-196; tinyobj attrib_t::texcoords, index_t::texcoord_index, mesh_t::material_ids).  The next
 * rt_build_lbvh (or this call, if the LBVH is already built) reorders them into BVH order next to triPos / triNor:
 * one 32-byte record per triangle {u0 v0 u1 v1 u2 v2 bits(material id) 0}.  material_ids may be NULL (all -1). */
int rt_set_surface(rt_ctx* ctx, const float* texcoords, uint32_t n_tex, const uint32_t* tex_idx /* [3T] */,
                   const int32_t* material_ids /* [T] or NULL */);
int rt_get_tri_uv(rt_ctx* ctx, float* tri_uv /* [8T], BVH order */);
/* For imported BVHs (rt_set_bvh / rt_load_bvh / rt_bvh_adopt_device): the records directly, in that BVH's order. */
int rt_set_tri_uv(rt_ctx* ctx, const float* tri_uv, uint32_t n_tri);
int rt_set_materials(rt_ctx* ctx, const rt_material_rec* materials, uint32_t n_materials);
/* Raw 8-bit RGB, row 0 = top of the image, rows tightly packed.  Decoding image files is the host's business. */
int rt_set_texture(rt_ctx* ctx, uint32_t slot, const uint8_t* rgb8, uint32_t width, uint32_t height);
/* Multi-GPU: the BVH-order records as device memory (for the host's RCCL broadcast), and their adoption (copied). */
int rt_tri_uv_device(rt_ctx* ctx, void** d_tri_uv, uint64_t* bytes);   /* NULL / 0 when the current BVH has none */
int rt_tri_uv_export_device(rt_ctx* ctx, void* d_dst, uint64_t dst_bytes /* >= 32 T */);   /* synchronous D2D copy */
int rt_adopt_tri_uv_device(rt_ctx* ctx, const void* d_tri_uv, uint32_t n_tri);

/* ------------------------------------------------------------------ multi-GPU: BVH replication (host drives RCCL) */
/* The flat buffers of a built/imported BVH as device memory, for ncclBroadcast by the host. */
typedef struct rt_bvh_view {
  void*    d_nodes;    uint64_t nodes_bytes;     /* 32 B x N */
  void*    d_tri_pos;  uint64_t tri_pos_bytes;   /* 48 B x T */
  void*    d_tri_nor;  uint64_t tri_nor_bytes;   /* 48 B x T */
  uint32_t n_nodes, n_tri;
} rt_bvh_view;
int rt_bvh_view_get(rt_ctx* ctx, rt_bvh_view* out);
/* Copy the three flat buffers into caller-owned device memory of the same sizes (e.g. buffers
 * that RCCL then broadcasts). Synchronous on the context stream. */
int rt_bvh_export_device(rt_ctx* ctx, const rt_bvh_view* dst);
/* Adopt caller-owned device buffers (filled by a broadcast) as this context's BVH and derive
 * the traversal layout from them.  The caller keeps the three buffers alive. */
int rt_bvh_adopt_device(rt_ctx* ctx, const rt_bvh_view* view);

/* Instrumentation for asynchronous frame loops (the counterpart of the reference's FPS counter,
 * script.js:951-957, at kernel resolution): between begin and end every rt_render_device records
 * hipEvents around the frame's kernels without synchronising; end synchronises the stream and
 * returns the averages over the (last <= 256) frames: ms_frame = cull + compact + traversal + stats,
 * ms_traverse = the traversal kernel alone. */
int rt_profile_begin(rt_ctx* ctx);
int rt_profile_end(rt_ctx* ctx, float* ms_frame_avg, float* ms_traverse_avg, uint32_t* n_frames);
/* The same with the traversal time by launch: rt_render_batch_device runs RT_MODE_PRIMARY / SHADOW pipelines as two traversal
 * launches - the primary rays' walk, then the colour of the pixels that hit something - and ms_traverse_avg / ms_shade_avg are
 * their two durations (ms_shade_avg = 0 where a pipeline has one traversal launch; rt_profile_end's figure is their sum). */
int rt_profile_end_phases(rt_ctx* ctx, float* ms_frame_avg, float* ms_traverse_avg, float* ms_shade_avg, uint32_t* n_frames);

/* Frame-coherent scheduling (on by default; no counterpart in the reference).  Every frame leaves one cost class per
 * pixel (how many traversal steps its rays took); the next frame of the same shape traces last frame's long pixels
 * first and packs pixels of one class into the same wavefronts.  The classes are hints: they change the ORDER in which
 * pixels are traced, never a pixel, a counter or a hit id — a frame with stale or no hints (first frame, new resolution,
 * new BVH, camera cut) is correct and merely runs at the unhinted speed.  Off: every frame is traced in tile order. */
int rt_set_coherent_scheduling(rt_ctx* ctx, int on);

/* Frames in flight (no counterpart in the reference, whose rAF loop submits one frame at a time, script.js:865-974).
 * A persistent traversal launch ends with a tail in which a few waves finish their last pixels (each pixel is a
 * chain of dependent node fetches), and small frames - a 1/8 stripe share of a 4K frame - are mostly tail.
 * With n = 2 .. 4, rt_render_device calls that do not ask for stats rotate over n internal frame streams (own
 * scratch each), so the tail of one frame overlaps the head of the next (small frames - a stripe share - profit from more).  Ordering rules in that mode:
 *   - a frame starts after everything queued on rt_stream() before its rt_render_device call;
 *   - rt_stream() itself does NOT wait for such frames until the next entry point that needs their results on it
 *     (rt_synchronize, any build / import / readback call, a rt_render* that returns stats);
 *   - rt_deinterleave_device never waits for frames (it consumes gathered data, not this context's frames);
 *   - rt_frames_join(ctx, keep_latest) makes rt_stream() wait for the frames issued so far, except the keep_latest most
 *     recent ones: a host that queues frame i first and then, on rt_stream(), the collective and the
 *     de-interleave of frame i-1 keeps two frames overlapping (raytracer_amd/dist.py does exactly that).
 * n = 1 (default): every frame runs on rt_stream(), in order. */
int rt_set_frames_in_flight(rt_ctx* ctx, int n /* 1 .. 4 */);
int rt_frames_join(rt_ctx* ctx, int keep_latest /* 0 .. 3 */);

int rt_synchronize(rt_ctx* ctx);
/* The hipStream_t the context launches on (as void*), so hosts can order their own work. */
void* rt_stream(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
