// rt_api.hip — the C ABI of include/rtcore.h on top of the gfx950 kernels.
// One context = one GPU + one HIP stream.  No CPU fallback exists: every compute entry
// point needs a usable HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_internal.h"

#include <atomic>
namespace {
std::string g_create_error;
std::atomic<unsigned long long> g_dev_alloc_calls{0};   // hipMalloc + hipFree calls of this library (every one goes through DevBuf)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) { g_dev_alloc_calls += 1; (void)hipFree(p); p = nullptr; n = 0; }
    if (count == 0) return hipSuccess;
    g_dev_alloc_calls += 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() { if (p) { g_dev_alloc_calls += 1; (void)hipFree(p); } p = nullptr; n = 0; }
};
}  // namespace

// ---------------------------------------------------------------- rt_render_device_noalloc: static device scratch
// The frame pipeline's scratch (compacted pixel list, per-tile class masks, per-block totals, queue heads, per-block
// counters) for callers that cannot allocate: one file-scope device array per GPU, held until the process ends.  Host
// code never uses the array's NAME as an address (it is a host-side stand-in): rt_create asks the runtime for the
// device address once per device (hipGetSymbolAddress, with that device current) and everything else - kernel
// arguments, hipMemsetAsync - uses that pointer.
namespace {
constexpr uint32_t NOALLOC_MAX_PIXELS = 3840u * 2160u;                 // one 4K frame (or stripe share) per call
constexpr uint32_t NOALLOC_MAX_TILES = NOALLOC_MAX_PIXELS / 64u + 16384u;   // 8x8 tiles incl. ragged right / bottom edges
constexpr uint32_t NOALLOC_MAX_GRID = 4096u;                           // >= rt_render_grid(n_cu) blocks (checked)
constexpr size_t NOALLOC_LIST_BYTES = sizeof(uint32_t) * (size_t)NOALLOC_MAX_PIXELS;
constexpr size_t NOALLOC_STATS_BYTES = sizeof(unsigned long long) * (8 + 32 + (size_t)NOALLOC_MAX_GRID * 8 + (size_t)RT_N_CLASS * NOALLOC_MAX_TILES);
constexpr size_t NOALLOC_TC_BYTES = sizeof(uint32_t) * (2 * 4096 + 32 + 64 + (size_t)RT_N_CLASS * (NOALLOC_MAX_TILES / 64 + 1));
constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t NOALLOC_OFF_STATS = align256(NOALLOC_LIST_BYTES);
constexpr size_t NOALLOC_OFF_TC = NOALLOC_OFF_STATS + align256(NOALLOC_STATS_BYTES);
constexpr size_t NOALLOC_BYTES = NOALLOC_OFF_TC + align256(NOALLOC_TC_BYTES);      // ~38.6 MB
}  // namespace
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];
namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};                // device address of `scratch` on each GPU (its own copy, its own address)
std::mutex g_scratch_mu;
// Called by rt_create with `dev` current.  Once per device.
hipError_t noalloc_scratch_init(int dev, unsigned char** out) {
  *out = nullptr;
  if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return hipSuccess;       // such a device simply has no no-alloc entry point
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    unsigned char* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return hipSuccess;
}
}  // namespace

struct rt_ctx {
  int device = 0;
  int n_cu = 256;
  unsigned char* noalloc_scratch = nullptr;   // this GPU's copy of the static scratch array (rt_render_device_noalloc)
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};
  std::string err;

  // rt_profile_begin/end: ring of event quadruples {frame begin, k_render begin, k_render end, frame end}
  static constexpr uint32_t PROF_RING = 256;
  std::vector<hipEvent_t> prof_ev;
  bool prof_on = false;
  uint32_t prof_n = 0;

  // mesh (loadObj contract)
  DevBuf<float> pos, nor;
  DevBuf<uint32_t> vidx, nidx;
  uint32_t n_vert = 0, n_nor = 0, n_tri_mesh = 0;

  // build scratch
  DevBuf<uint32_t> bounds_enc, codes, keys_a, keys_b, vals_a, vals_b, hist, tile_status, tile_counter, words;
  DevBuf<float> bounds_f, centroids;
  DevBuf<int32_t> span, split;
  DevBuf<float4> mesh_rec;
  DevBuf<float> tri_in_pos;
  DevBuf<uint32_t> cross_list;
  DevBuf<float4> refit_tables;
  bool built = false;      // codes / order valid
  uint32_t refit_fallbacks = 0;

  // BVH
  RtBvhDev bvh;
  DevBuf<float> own_nodes, own_tri_pos, own_tri_nor;
  DevBuf<float4> wide, tri_pre;
  bool build_stage_times = false;        // rt_set_build_stage_timing
  DevBuf<float4> wide_paired;            // experiment (RTCORE_WIDE_PAIRED): see rt_launch_pair_wide
  bool have_paired = false;
  bool have_bvh = false;
  int node_encoding = RT_NODES_V1;   // encoding of the next built / imported / adopted node buffer

  // surface extension (rtcore.h "material / texture shading")
  DevBuf<float> texcoords, tri_uv;
  DevBuf<uint32_t> tidx;
  DevBuf<int32_t> matid;
  bool surface_set = false, surface_has_mat = false;   // mesh-order arrays uploaded for the current mesh
  uint32_t tri_uv_n = 0;                                // triangles described by tri_uv (0 = none)
  DevBuf<float4> materials;
  uint32_t n_materials = 0;
  std::vector<rt_material_rec> materials_host;
  DevBuf<uint32_t> texels[RT_MAX_TEXTURES];
  RtTexDev tex_host[RT_MAX_TEXTURES] = {};
  DevBuf<RtTexDev> tex_table;
  uint32_t n_textures = 0;                              // highest set slot + 1

  // render scratch, one set per frame slot (slot 0 = the context stream; slots 0/1 alternate with two frames in flight)
  static constexpr int MAX_FIF = 4;            // frame slots (rt_set_frames_in_flight)
  DevBuf<uint32_t> r_tile_counter[MAX_FIF], r_list[MAX_FIF];
  DevBuf<uint32_t> r_hits[MAX_FIF];            // traverse / shade split: the primary hit's triangle per list entry
  DevBuf<RtFrameRec> r_frames[MAX_FIF];        // a batch's per-frame records (rt_internal.h)
  DevBuf<float4> pt_recs[MAX_FIF];             // PT split records (rt_trace.hip), per frame slot
  DevBuf<float> pt_terms[MAX_FIF];
  DevBuf<unsigned long long> r_stats[MAX_FIF];
  DevBuf<unsigned long long> r_bstats[MAX_FIF];   // a counted batch: [RT_MAX_BATCH][9] totals, then [n][resident waves][8] per-wave partials
  int bstats_slot = -1;                           // frame slot + frame count of the last batch rendered with RT_FLAG_COUNT_WORK (rt_batch_stats)
  uint32_t bstats_n = 0;
  // frames in flight (rt_set_frames_in_flight): asynchronous frames alternate between two frame streams so that the
  // tail of frame i (few waves left, latency-bound) overlaps the head of frame i+1
  DevBuf<uint8_t> pix_class;                   // frame-coherent scheduling hints (rt_trace.hip), shared by the frame slots
  uint64_t pix_class_key = ~0ull;
  float last_uniforms[44] = {};                // view of the previous frame: an unchanged view keeps per-pixel hints unpromoted
  bool class_partial_reset[MAX_FIF] = {true, true, true, true};
  uint32_t bvh_serial = 0;                     // bumped whenever the BVH changes
  bool coherent_scheduling = true;
  int frames_in_flight = 1;
  hipStream_t fstream[MAX_FIF] = {};
  hipEvent_t f_done[MAX_FIF] = {};             // recorded behind each async frame on its frame stream
  hipEvent_t f_main = nullptr;                 // main-stream marker the frame streams wait on
  bool f_pending[MAX_FIF] = {};                // f_done[k] recorded and not yet joined into the context stream
  uint32_t f_seq[MAX_FIF] = {};                // issue number of the frame pending in slot k
  uint32_t f_count = 0;
  DevBuf<uint8_t> o_rgba;
  DevBuf<uint32_t> o_tri;
  DevBuf<float> o_t;

  int fail(int code, const std::string& msg) { err = msg; return code; }
  int hipfail(const char* what, hipError_t e) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_HIP;
  }
};

#define HIPCHK(expr)                                             \
  do {                                                           \
    hipError_t e_ = (expr);                                      \
    if (e_ != hipSuccess) return ctx->hipfail(#expr, e_);        \
  } while (0)

namespace {
struct DeviceGuard {   // makes the context's device current for the duration of a call
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Make the context stream wait for every asynchronous frame still running on a frame stream.
int join_frames(rt_ctx* ctx) {
  for (int k = 0; k < rt_ctx::MAX_FIF; ++k)
    if (ctx->f_pending[k]) {
      hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->f_done[k], 0);
      if (e != hipSuccess) return ctx->hipfail("hipStreamWaitEvent(frame)", e);
      ctx->f_pending[k] = false;
    }
  return RT_OK;
}
#define JOIN_FRAMES() do { const int jrc_ = join_frames(ctx); if (jrc_ != RT_OK) return jrc_; } while (0)

// A frame (or batch) that rotates over the frame streams takes its slot's issue number before work that can still fail
// (scratch growth, limits): unless the pipeline was really queued (done = true), the number and the slot's previous issue
// number come back, so rt_frames_join's keep_latest accounting and the slot rotation stay exact.
struct SlotTicket {
  rt_ctx* c; int slot = 0; uint32_t old_seq = 0; bool taken = false, done = false;
  explicit SlotTicket(rt_ctx* ctx) : c(ctx) {}
  void take(int k) { slot = k; old_seq = c->f_seq[k]; c->f_seq[k] = c->f_count++; taken = true; }
  ~SlotTicket() { if (taken && !done) { c->f_count -= 1; c->f_seq[slot] = old_seq; } }
};

// limits of the two node-record encodings
int check_encoding_limits(rt_ctx* ctx, const char* who, uint64_t n_nodes, uint64_t n_tri) {
  if (ctx->node_encoding == RT_NODES_V1 && n_nodes > (1ull << 24))
    return ctx->fail(RT_ERR_LIMIT, std::string(who) + ": v1 node records store indices as f32 values, exact only below 2^24 nodes "
                                   "(SURVEY F11); select RT_NODES_V2 with rt_set_node_encoding");
  if (n_tri > (1ull << 28) || n_nodes > (1ull << 29))
    return ctx->fail(RT_ERR_LIMIT, std::string(who) + ": more than 2^28 triangles");
  return RT_OK;
}

// WIDE64 leaf ref = start | (count-1) << leaf_shift in 31 bits: 23 keeps counts up to 256 for scenes below
// 2^23 triangles; larger scenes give the start more bits (count bits = 31 - leaf_shift)
int leaf_shift_for(uint64_t n_tri) {
  int sh = 23;
  while (sh < 31 && n_tri > (1ull << sh)) sh += 1;
  return sh;
}

// Derive the traversal layout from the interchange buffers (imports), or take the one the refit already wrote (builds).
int finish_bvh(rt_ctx* ctx, bool packed_by_refit = false) {
  RtBvhDev& b = ctx->bvh;
  b.v2 = ctx->node_encoding == RT_NODES_V2;
  b.leaf_shift = leaf_shift_for(b.n_tri);
  b.n_wide = b.n_nodes;   // imported trees may number internal nodes anywhere in [0, N)
  HIPCHK(ctx->wide.ensure((size_t)b.n_wide * 4));
  HIPCHK(ctx->tri_pre.ensure((size_t)b.n_tri * 3));
  HIPCHK(ctx->words.ensure(4));
  b.wide = ctx->wide.p; b.tri_pre = ctx->tri_pre.p;
  uint32_t w[2] = {0, 0};
  if (packed_by_refit) {   // Karras numbering: root = node 0, internal unless the scene is a single triangle; root_box was fetched by the build
    w[0] = (b.n_tri == 1) ? 0x80000000u : 0u;
  } else {
    HIPCHK(hipMemsetAsync(ctx->words.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    rt_launch_pack_impl(b, ctx->words.p + 0, ctx->words.p + 1, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(w, ctx->words.p, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(b.root_box, b.nodes, sizeof(b.root_box), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (w[1] & 2u)
    return ctx->fail(RT_ERR_LIMIT, "a leaf's start/count does not fit the WIDE64 leaf reference (start < 2^" + std::to_string(b.leaf_shift) +
                                   ", count <= " + std::to_string(b.leaf_shift >= 31 ? 1u : (1u << (31 - b.leaf_shift))) + ")");
  if (w[1] & 4u) return ctx->fail(RT_ERR_INVALID, "node buffer has an out-of-range child index or a leaf range outside the triangle buffer");
  b.root_ref = w[0];
  ctx->have_bvh = true;
  ctx->bvh_serial += 1;
  return RT_OK;
}
}  // namespace

// ================================================================ context
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION; }
extern "C" uint64_t rt_debug_device_allocations(void) { return g_dev_alloc_calls.load(); }

extern "C" int rt_create(rt_ctx** out, const char* backend, const int* device_ids, int n_devices) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  if (backend && std::strcmp(backend, "hip") != 0) {
    g_create_error = "only backend \"hip\" exists (the CPU oracle is test infrastructure, not a backend)";
    return RT_ERR_INVALID;
  }
  if (n_devices > 1) { g_create_error = "one context drives one GPU; create one context (process) per device"; return RT_ERR_INVALID; }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return RT_ERR_NO_DEVICE;
  }
  const int dev = (device_ids && n_devices == 1) ? device_ids[0] : 0;
  if (dev < 0 || dev >= count) { g_create_error = "device id out of range"; return RT_ERR_INVALID; }
  rt_ctx* ctx = new rt_ctx();
  ctx->device = dev;
  if ((e = hipSetDevice(dev)) != hipSuccess) { g_create_error = hipGetErrorString(e); delete ctx; return RT_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  if ((e = noalloc_scratch_init(dev, &ctx->noalloc_scratch)) != hipSuccess) {
    g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); delete ctx; return RT_ERR_HIP;
  }
  if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) { g_create_error = hipGetErrorString(e); delete ctx; return RT_ERR_HIP; }
  for (auto& ev : ctx->ev) (void)hipEventCreate(&ev);
  if (const char* e = std::getenv("RTCORE_COHERENT_SCHEDULING")) ctx->coherent_scheduling = std::atoi(e) != 0;
  *out = ctx;
  return RT_OK;
}

extern "C" void rt_destroy(rt_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->pos.release(); ctx->nor.release(); ctx->vidx.release(); ctx->nidx.release();
  ctx->bounds_enc.release(); ctx->codes.release(); ctx->keys_a.release(); ctx->keys_b.release();
  ctx->vals_a.release(); ctx->vals_b.release(); ctx->hist.release(); ctx->tile_status.release();
  ctx->tile_counter.release(); ctx->words.release(); ctx->bounds_f.release(); ctx->centroids.release();
  ctx->span.release(); ctx->split.release(); ctx->mesh_rec.release(); ctx->tri_in_pos.release(); ctx->cross_list.release(); ctx->refit_tables.release();
  ctx->own_nodes.release(); ctx->own_tri_pos.release(); ctx->own_tri_nor.release();
  ctx->wide.release(); ctx->tri_pre.release(); ctx->wide_paired.release();
  ctx->texcoords.release(); ctx->tri_uv.release(); ctx->tidx.release(); ctx->matid.release(); ctx->materials.release();
  for (auto& t : ctx->texels) t.release();
  ctx->tex_table.release();
  for (int k = 0; k < rt_ctx::MAX_FIF; ++k) {
    ctx->r_tile_counter[k].release(); ctx->r_list[k].release(); ctx->r_hits[k].release(); ctx->r_frames[k].release(); ctx->r_stats[k].release(); ctx->r_bstats[k].release(); ctx->pt_recs[k].release(); ctx->pt_terms[k].release();
    if (ctx->fstream[k]) { (void)hipStreamSynchronize(ctx->fstream[k]); (void)hipStreamDestroy(ctx->fstream[k]); }
    if (ctx->f_done[k]) (void)hipEventDestroy(ctx->f_done[k]);
  }
  if (ctx->f_main) (void)hipEventDestroy(ctx->f_main);
  ctx->o_rgba.release(); ctx->o_tri.release(); ctx->o_t.release(); ctx->pix_class.release();
  for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->prof_ev) if (ev) (void)hipEventDestroy(ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
extern "C" int rt_synchronize(rt_ctx* ctx) {
  if (!ctx) return RT_ERR_INVALID;
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}
extern "C" void* rt_stream(rt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ================================================================ hot path 1
extern "C" int rt_set_node_encoding(rt_ctx* ctx, int encoding) {
  if (!ctx) return RT_ERR_INVALID;
  if (encoding != RT_NODES_V1 && encoding != RT_NODES_V2) return ctx->fail(RT_ERR_INVALID, "rt_set_node_encoding: unknown encoding");
  ctx->node_encoding = encoding;
  return RT_OK;
}
extern "C" int rt_get_node_encoding(rt_ctx* ctx) { return ctx ? (ctx->have_bvh ? (ctx->bvh.v2 ? RT_NODES_V2 : RT_NODES_V1) : ctx->node_encoding) : RT_ERR_INVALID; }

extern "C" int rt_set_mesh(rt_ctx* ctx, const float* positions, uint32_t n_vert, const float* normals, uint32_t n_nor,
                           const uint32_t* vert_idx, const uint32_t* nor_idx, uint32_t n_tri) {
  if (!ctx) return RT_ERR_INVALID;
  if (!positions || !normals || !vert_idx || !nor_idx || n_vert == 0 || n_nor == 0 || n_tri == 0)
    return ctx->fail(RT_ERR_INVALID, "rt_set_mesh: empty mesh or NULL array");
  if (n_tri > (1u << 30)) return ctx->fail(RT_ERR_LIMIT, "rt_set_mesh: more than 2^30 triangles");
  for (size_t i = 0; i < (size_t)n_tri * 3; ++i)   // a wild index would fault the GPU: check on the host
    if (vert_idx[i] >= n_vert || nor_idx[i] >= n_nor) return ctx->fail(RT_ERR_INVALID, "rt_set_mesh: index out of range");
  // NaN / Inf positions have no defined Morton code or box order (WGSL leaves clamp / min / max of NaN open): out of contract
  for (size_t i = 0; i < (size_t)n_vert * 3; ++i)
    if (!(positions[i] - positions[i] == 0.0f)) return ctx->fail(RT_ERR_INVALID, "rt_set_mesh: non-finite vertex position");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(ctx->pos.ensure((size_t)n_vert * 3));
  HIPCHK(ctx->nor.ensure((size_t)n_nor * 3));
  HIPCHK(ctx->vidx.ensure((size_t)n_tri * 3));
  HIPCHK(ctx->nidx.ensure((size_t)n_tri * 3));
  HIPCHK(hipMemcpyAsync(ctx->pos.p, positions, sizeof(float) * 3 * n_vert, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->nor.p, normals, sizeof(float) * 3 * n_nor, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->vidx.p, vert_idx, sizeof(uint32_t) * 3 * n_tri, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->nidx.p, nor_idx, sizeof(uint32_t) * 3 * n_tri, hipMemcpyHostToDevice, ctx->stream));
  // device layout of the mesh: de-indexed once, one 128-byte record per triangle (rt_build.hip k_pack_mesh); the records'
  // texcoord / material slots are filled by a later rt_set_surface
  HIPCHK(ctx->mesh_rec.ensure(rt_mesh_rec_float4s(n_tri)));
  HIPCHK(ctx->tri_in_pos.ensure((size_t)n_tri * 9));
  rt_launch_pack_mesh(ctx->pos.p, ctx->nor.p, ctx->vidx.p, ctx->nidx.p, nullptr, nullptr, nullptr, n_tri, ctx->mesh_rec.p, ctx->tri_in_pos.p, ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_vert = n_vert; ctx->n_nor = n_nor; ctx->n_tri_mesh = n_tri;
  ctx->built = false;
  ctx->surface_set = false;      // texcoords / material ids belong to the previous mesh
  return RT_OK;
}

extern "C" int rt_set_build_stage_timing(rt_ctx* ctx, int on) {
  if (!ctx) return RT_ERR_INVALID;
  ctx->build_stage_times = on != 0;
  return RT_OK;
}

extern "C" int rt_build_lbvh(rt_ctx* ctx, rt_build_stats* out) {
  if (!ctx) return RT_ERR_INVALID;
  if (ctx->n_tri_mesh == 0) return ctx->fail(RT_ERR_INVALID, "rt_build_lbvh: call rt_set_mesh first");
  const uint32_t T = ctx->n_tri_mesh, N = 2 * T - 1;
  { const int lrc = check_encoding_limits(ctx, "rt_build_lbvh", N, T); if (lrc != RT_OK) return lrc; }
  const bool v2 = ctx->node_encoding == RT_NODES_V2;
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  // the buffers of the previous BVH are reallocated / overwritten below: until this build has succeeded there is no BVH
  // (an early return must not leave have_bvh pointing at freed or half-written records)
  ctx->have_bvh = false; ctx->built = false; ctx->tri_uv_n = 0; ctx->have_paired = false;
  hipStream_t s = ctx->stream;
  HIPCHK(ctx->bounds_enc.ensure(rt_bounds_words())); HIPCHK(ctx->bounds_f.ensure(8)); HIPCHK(ctx->centroids.ensure((size_t)T * 3));
  HIPCHK(ctx->codes.ensure(T)); HIPCHK(ctx->keys_a.ensure(T)); HIPCHK(ctx->keys_b.ensure(T));
  HIPCHK(ctx->vals_a.ensure(T)); HIPCHK(ctx->vals_b.ensure(T));
  HIPCHK(ctx->hist.ensure(rt_sort_hist_words())); HIPCHK(ctx->tile_status.ensure(rt_sort_status_words(T)));
  HIPCHK(ctx->tile_counter.ensure(rt_build_counter_words())); HIPCHK(ctx->words.ensure(4));
  HIPCHK(ctx->span.ensure(T > 1 ? T - 1 : 1)); HIPCHK(ctx->split.ensure(T > 1 ? T - 1 : 1));
  HIPCHK(ctx->cross_list.ensure((size_t)64 * rt_cross_capacity(T))); HIPCHK(ctx->refit_tables.ensure(rt_refit_table_float4s(T)));
  HIPCHK(ctx->own_nodes.ensure((size_t)N * 8)); HIPCHK(ctx->own_tri_pos.ensure((size_t)T * 12));
  HIPCHK(ctx->own_tri_nor.ensure((size_t)T * 12));
  HIPCHK(ctx->wide.ensure((size_t)N * 4)); HIPCHK(ctx->tri_pre.ensure((size_t)T * 3));

  RtBuildBuffers b;
  b.pos = ctx->pos.p; b.nor = ctx->nor.p; b.vidx = ctx->vidx.p; b.nidx = ctx->nidx.p; b.n_tri = T;
  b.bounds_enc = ctx->bounds_enc.p; b.bounds_f = ctx->bounds_f.p; b.centroids = ctx->centroids.p; b.codes = ctx->codes.p;
  b.keys_a = ctx->keys_a.p; b.keys_b = ctx->keys_b.p; b.vals_a = ctx->vals_a.p; b.vals_b = ctx->vals_b.p;
  b.hist = ctx->hist.p; b.tile_status = ctx->tile_status.p; b.tile_counter = ctx->tile_counter.p;
  b.err_flag = ctx->words.p + 2; b.span = ctx->span.p; b.split = ctx->split.p; b.mesh_rec = ctx->mesh_rec.p; b.tri_in_pos = ctx->tri_in_pos.p; b.cross_list = ctx->cross_list.p; b.refit_tables = ctx->refit_tables.p;
  b.n_sort_tiles = rt_sort_tile_count(T);
  b.texcoords = nullptr; b.tex_idx = nullptr; b.material_ids = nullptr; b.tri_uv = nullptr;
  if (ctx->surface_set) {          // surface records into BVH order by the refit, next to triPos / triNor
    HIPCHK(ctx->tri_uv.ensure((size_t)T * 8));
    b.texcoords = ctx->texcoords.p; b.tex_idx = ctx->tidx.p; b.material_ids = ctx->surface_has_mat ? ctx->matid.p : nullptr;
    b.tri_uv = reinterpret_cast<float4*>(ctx->tri_uv.p);
  }

  HIPCHK(hipMemsetAsync(ctx->words.p, 0, 4 * sizeof(uint32_t), s));
  // per-stage hipEvents sit between the build's launches and cost it ~2 us each (0.279 -> 0.288 ms at 1.31 M triangles):
  // recorded only on request (rt_set_build_stage_timing / RTCORE_BUILD_STAGE_TIMES=1), ms_total is always first launch -> last
  bool stage_events = ctx->build_stage_times;
  if (const char* e = getenv("RTCORE_BUILD_STAGE_TIMES")) stage_events = e[0] == '1';
  HIPCHK(hipEventRecord(ctx->ev[0], s));
  rt_launch_bounds(b, s);
  if (stage_events) HIPCHK(hipEventRecord(ctx->ev[1], s));
  rt_launch_morton(b, s);
  if (stage_events) HIPCHK(hipEventRecord(ctx->ev[2], s));
  rt_launch_sort(b, s);
  if (stage_events) HIPCHK(hipEventRecord(ctx->ev[3], s));
  rt_launch_refit_local(b, ctx->own_nodes.p, ctx->own_tri_pos.p, ctx->own_tri_nor.p, ctx->wide.p, ctx->tri_pre.p, v2, s);
  if (stage_events) HIPCHK(hipEventRecord(ctx->ev[4], s));
  rt_launch_refit_upper(b, ctx->own_nodes.p, ctx->wide.p, v2, s);
  HIPCHK(hipEventRecord(ctx->ev[5], s));
  HIPCHK(hipGetLastError());
  // one read-back for the whole build: the sort's error word and the root record (the frame pipeline's screen-space
  // cull rectangle is computed on the host from it)
  uint32_t errw = 0;
  float root_box[8];
  HIPCHK(hipMemcpyAsync(&errw, ctx->words.p + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(root_box, ctx->own_nodes.p, sizeof(root_box), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (errw & 1u) return ctx->fail(RT_ERR_INTERNAL, "rt_build_lbvh: radix-sort look-back spin limit expired");

  ctx->bvh = RtBvhDev();
  ctx->bvh.nodes = ctx->own_nodes.p; ctx->bvh.tri_pos = ctx->own_tri_pos.p; ctx->bvh.tri_nor = ctx->own_tri_nor.p;
  ctx->bvh.n_nodes = N; ctx->bvh.n_tri = T; ctx->bvh.owns_interchange = true;
  std::memcpy(ctx->bvh.root_box, root_box, sizeof(root_box));
  HIPCHK(hipEventRecord(ctx->ev[6], s));
  ctx->tri_uv_n = ctx->surface_set ? T : 0;
  int rc = finish_bvh(ctx, /*packed_by_refit=*/true);
  if (rc != RT_OK) return rc;
  if (getenv("RTCORE_WIDE_PAIRED") && T > 2) {      // experiment: built when the variable exists, used when it is "1" (rt_render*)
    const uint32_t n_int = T - 1;
    HIPCHK(ctx->wide_paired.ensure((size_t)2 * n_int * 4));
    HIPCHK(hipMemsetAsync(ctx->words.p, 0, 4 * sizeof(uint32_t), s));
    // up / pos borrow the sort's key buffers (free once the refit is done; T words each)
    rt_launch_pair_wide(ctx->wide.p, n_int, ctx->keys_b.p, ctx->vals_b.p, ctx->words.p, ctx->wide_paired.p, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    ctx->have_paired = true;
  }
  HIPCHK(hipEventRecord(ctx->ev[7], s));
  HIPCHK(hipEventSynchronize(ctx->ev[7]));
  ctx->built = true;
  if (out) {
    std::memset(out, 0, sizeof(*out));
    if (stage_events) {
    (void)hipEventElapsedTime(&out->ms_bounds, ctx->ev[0], ctx->ev[1]);
    (void)hipEventElapsedTime(&out->ms_morton, ctx->ev[1], ctx->ev[2]);
    (void)hipEventElapsedTime(&out->ms_sort, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&out->ms_refit, ctx->ev[3], ctx->ev[4]);   // leaves + the tree and boxes inside every 256-leaf block
    (void)hipEventElapsedTime(&out->ms_tree, ctx->ev[4], ctx->ev[5]);    // the nodes above the blocks
    }
    if (ctx->have_paired) (void)hipEventElapsedTime(&out->ms_pack, ctx->ev[6], ctx->ev[7]);   // else 0: a build launches nothing here (two events on an idle stream read ~4 us apart)
    out->ms_total = out->ms_bounds + out->ms_morton + out->ms_sort + out->ms_tree + out->ms_refit + out->ms_pack;
    if (!stage_events) { float t = 0.0f; (void)hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[5]); out->ms_total = t + out->ms_pack; }   // the stage fields stay 0
    out->n_tri = T; out->n_nodes = N;
    out->algorithmic_bytes = 560ull * T;
    out->refit_fallbacks = ctx->refit_fallbacks;
  }
  return RT_OK;
}

#define NEED_BUILT()  if (!ctx) return RT_ERR_INVALID; if (!ctx->built) return ctx->fail(RT_ERR_INVALID, "no LBVH has been built in this context")
#define NEED_BVH()    if (!ctx) return RT_ERR_INVALID; if (!ctx->have_bvh) return ctx->fail(RT_ERR_INVALID, "no BVH: call rt_build_lbvh / rt_set_bvh / rt_bvh_adopt_device")

extern "C" int rt_get_bounds(rt_ctx* ctx, float bmin[3], float inv_extent[3]) {
  NEED_BUILT();
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  float f[8];
  HIPCHK(hipMemcpy(f, ctx->bounds_f.p, sizeof(f), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; ++k) { bmin[k] = f[k]; inv_extent[k] = f[4 + k]; }
  return RT_OK;
}
extern "C" int rt_get_morton(rt_ctx* ctx, uint32_t* codes) {
  NEED_BUILT();
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipMemcpy(codes, ctx->codes.p, sizeof(uint32_t) * ctx->bvh.n_tri, hipMemcpyDeviceToHost));
  return RT_OK;
}
extern "C" int rt_get_tri_order(rt_ctx* ctx, uint32_t* order) {
  NEED_BUILT();
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipMemcpy(order, ctx->vals_a.p, sizeof(uint32_t) * ctx->bvh.n_tri, hipMemcpyDeviceToHost));
  return RT_OK;
}
extern "C" int rt_get_counts(rt_ctx* ctx, uint32_t* n_tri, uint32_t* n_nodes) {
  NEED_BVH();
  if (n_tri) *n_tri = ctx->bvh.n_tri;
  if (n_nodes) *n_nodes = ctx->bvh.n_nodes;
  return RT_OK;
}
extern "C" int rt_get_nodes(rt_ctx* ctx, float* nodes) {
  NEED_BVH();
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipMemcpy(nodes, ctx->bvh.nodes, sizeof(float) * 8 * (size_t)ctx->bvh.n_nodes, hipMemcpyDeviceToHost));
  return RT_OK;
}
extern "C" int rt_get_triangles(rt_ctx* ctx, float* tri_pos, float* tri_nor) {
  NEED_BVH();
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  if (tri_pos) HIPCHK(hipMemcpy(tri_pos, ctx->bvh.tri_pos, sizeof(float) * 12 * (size_t)ctx->bvh.n_tri, hipMemcpyDeviceToHost));
  if (tri_nor) HIPCHK(hipMemcpy(tri_nor, ctx->bvh.tri_nor, sizeof(float) * 12 * (size_t)ctx->bvh.n_tri, hipMemcpyDeviceToHost));
  return RT_OK;
}

namespace {
// host-side structural validation of an imported node buffer: a bad index would fault the GPU
int validate_nodes(rt_ctx* ctx, const float* nodes, uint32_t n_nodes, uint32_t n_tri) {
  if (ctx->node_encoding == RT_NODES_V2) {
    for (uint32_t i = 0; i < n_nodes; ++i) {
      uint32_t A, B;
      std::memcpy(&A, nodes + 8 * (size_t)i + 3, 4); std::memcpy(&B, nodes + 8 * (size_t)i + 7, 4);
      if (B >> 31) {
        const uint64_t cnt = B & 0x7FFFFFFFu;
        if (cnt == 0 || (uint64_t)A + cnt > n_tri)
          return ctx->fail(RT_ERR_INVALID, "imported BVH (v2): leaf range outside the triangle buffer at node " + std::to_string(i));
      } else if (A >= n_nodes || B >= n_nodes) {
        return ctx->fail(RT_ERR_INVALID, "imported BVH (v2): child index out of range at node " + std::to_string(i));
      }
    }
    return RT_OK;
  }
  for (uint32_t i = 0; i < n_nodes; ++i) {
    const float a = nodes[8 * (size_t)i + 3], b = nodes[8 * (size_t)i + 7];
    if (b < 0.0f) {
      const double start = a, cnt = -(double)b;
      if (!(start >= 0) || start != (double)(int64_t)start || cnt != (double)(int64_t)cnt || start + cnt > (double)n_tri)
        return ctx->fail(RT_ERR_INVALID, "imported BVH: leaf range outside the triangle buffer at node " + std::to_string(i));
    } else {
      if (!(a >= 0) || !(b >= 0) || a >= (double)n_nodes || b >= (double)n_nodes || a != (double)(int64_t)a || b != (double)(int64_t)b)
        return ctx->fail(RT_ERR_INVALID, "imported BVH: child index out of range at node " + std::to_string(i));
    }
  }
  return RT_OK;
}
}  // namespace

extern "C" int rt_set_bvh(rt_ctx* ctx, const float* nodes, uint32_t n_nodes, const float* tri_pos, const float* tri_nor, uint32_t n_tri) {
  if (!ctx) return RT_ERR_INVALID;
  if (!nodes || !tri_pos || !tri_nor || n_nodes == 0 || n_tri == 0) return ctx->fail(RT_ERR_INVALID, "rt_set_bvh: empty input");
  int rc = check_encoding_limits(ctx, "rt_set_bvh", n_nodes, n_tri);
  if (rc != RT_OK) return rc;
  rc = validate_nodes(ctx, nodes, n_nodes, n_tri);
  if (rc != RT_OK) return rc;
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  ctx->have_bvh = false; ctx->built = false; ctx->tri_uv_n = 0;
  HIPCHK(ctx->own_nodes.ensure((size_t)n_nodes * 8));
  HIPCHK(ctx->own_tri_pos.ensure((size_t)n_tri * 12));
  HIPCHK(ctx->own_tri_nor.ensure((size_t)n_tri * 12));
  HIPCHK(hipMemcpyAsync(ctx->own_nodes.p, nodes, sizeof(float) * 8 * (size_t)n_nodes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->own_tri_pos.p, tri_pos, sizeof(float) * 12 * (size_t)n_tri, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->own_tri_nor.p, tri_nor, sizeof(float) * 12 * (size_t)n_tri, hipMemcpyHostToDevice, ctx->stream));
  ctx->bvh = RtBvhDev();
  ctx->bvh.nodes = ctx->own_nodes.p; ctx->bvh.tri_pos = ctx->own_tri_pos.p; ctx->bvh.tri_nor = ctx->own_tri_nor.p;
  ctx->bvh.n_nodes = n_nodes; ctx->bvh.n_tri = n_tri;
  return finish_bvh(ctx);
}

// Cache-file trailer: 16 bytes {magic "RTBV", n_nodes, n_tri, node encoding} behind the three raw buffers.  Written only
// where the header-less file could not be read back: trees with N != 2T-1 (multi-triangle leaves, e.g. the reference's
// SAH builder with leaf > 1, script.js:285-513) and v2 node records.  A leaf-1 v1 tree stays the raw f32 dump that can be
// bound to the reference's buffers as it is.
namespace {
constexpr uint32_t BVH_FILE_MAGIC = 0x56425452u;   // "RTBV" little-endian
}

extern "C" int rt_save_bvh(rt_ctx* ctx, const char* path) {
  NEED_BVH();
  if (!path) return ctx->fail(RT_ERR_INVALID, "rt_save_bvh: NULL path");
  const size_t N = ctx->bvh.n_nodes, T = ctx->bvh.n_tri;
  std::vector<float> buf(8 * N + 24 * T);
  int rc = rt_get_nodes(ctx, buf.data());
  if (rc != RT_OK) return rc;
  rc = rt_get_triangles(ctx, buf.data() + 8 * N, buf.data() + 8 * N + 12 * T);
  if (rc != RT_OK) return rc;
  FILE* f = std::fopen(path, "wb");
  if (!f) return ctx->fail(RT_ERR_IO, std::string("cannot open for writing: ") + path);
  size_t w = std::fwrite(buf.data(), sizeof(float), buf.size(), f);
  bool ok = w == buf.size();
  if (ok && (N != 2 * T - 1 || ctx->bvh.v2)) {
    const uint32_t tr[4] = {BVH_FILE_MAGIC, (uint32_t)N, (uint32_t)T, (uint32_t)(ctx->bvh.v2 ? RT_NODES_V2 : RT_NODES_V1)};
    ok = std::fwrite(tr, sizeof(uint32_t), 4, f) == 4;
  }
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) return ctx->fail(RT_ERR_IO, std::string("short write: ") + path);
  return RT_OK;
}

extern "C" int rt_load_bvh(rt_ctx* ctx, const char* path) {
  if (!ctx) return RT_ERR_INVALID;
  if (!path) return ctx->fail(RT_ERR_INVALID, "rt_load_bvh: NULL path");
  FILE* f = std::fopen(path, "rb");
  if (!f) return ctx->fail(RT_ERR_IO, std::string("cannot open: ") + path);
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (sz <= 0 || sz % 4 != 0) { std::fclose(f); return ctx->fail(RT_ERR_IO, "not a BVH cache file (empty, or not a whole number of 32-bit words)"); }
  std::vector<float> buf((size_t)sz / 4);
  const size_t r = std::fread(buf.data(), 1, (size_t)sz, f);
  std::fclose(f);
  if (r != (size_t)sz) return ctx->fail(RT_ERR_IO, "short read");
  size_t T = 0, N = 0;
  int file_encoding = 0;
  if (sz >= 16 + 32 + 96) {                          // trailer {magic, N, T, encoding} consistent with the file size?
    uint32_t tr[4];
    std::memcpy(tr, buf.data() + buf.size() - 4, sizeof(tr));
    if (tr[0] == BVH_FILE_MAGIC && (tr[3] == RT_NODES_V1 || tr[3] == RT_NODES_V2) && tr[1] > 0 && tr[2] > 0 &&
        32ull * tr[1] + 96ull * tr[2] + 16ull == (unsigned long long)sz) {
      N = tr[1]; T = tr[2];
      file_encoding = (int)tr[3];                    // the file says how its slots A / B are encoded
    }
  }
  if (N == 0) {                                      // header-less: 32 B x (2T-1) + 96 B x T = 160 T - 32, encoding = the context's setting
    if ((sz + 32) % 160 != 0) return ctx->fail(RT_ERR_IO, "not a BVH cache file (no trailer, and the size is not 160*T-32 bytes)");
    T = (size_t)(sz + 32) / 160; N = 2 * T - 1;
  }
  // a trailer's encoding becomes the context's setting only if the tree is accepted: a rejected / corrupt file must not
  // change how the next rt_build_lbvh encodes its records
  const int old_encoding = ctx->node_encoding;
  if (file_encoding) ctx->node_encoding = file_encoding;
  const int rc = rt_set_bvh(ctx, buf.data(), (uint32_t)N, buf.data() + 8 * N, buf.data() + 8 * N + 12 * T, (uint32_t)T);
  if (rc != RT_OK) ctx->node_encoding = old_encoding;
  return rc;
}

// ================================================================ multi-GPU plumbing
extern "C" int rt_bvh_view_get(rt_ctx* ctx, rt_bvh_view* out) {
  NEED_BVH();
  if (!out) return RT_ERR_INVALID;
  out->d_nodes = ctx->bvh.nodes; out->nodes_bytes = 32ull * ctx->bvh.n_nodes;
  out->d_tri_pos = ctx->bvh.tri_pos; out->tri_pos_bytes = 48ull * ctx->bvh.n_tri;
  out->d_tri_nor = ctx->bvh.tri_nor; out->tri_nor_bytes = 48ull * ctx->bvh.n_tri;
  out->n_nodes = ctx->bvh.n_nodes; out->n_tri = ctx->bvh.n_tri;
  return RT_OK;
}

extern "C" int rt_bvh_export_device(rt_ctx* ctx, const rt_bvh_view* dst) {
  NEED_BVH();
  if (!dst || !dst->d_nodes || !dst->d_tri_pos || !dst->d_tri_nor) return ctx->fail(RT_ERR_INVALID, "rt_bvh_export_device: incomplete view");
  const uint64_t nb = 32ull * ctx->bvh.n_nodes, tb = 48ull * ctx->bvh.n_tri;
  if (dst->nodes_bytes < nb || dst->tri_pos_bytes < tb || dst->tri_nor_bytes < tb)
    return ctx->fail(RT_ERR_INVALID, "rt_bvh_export_device: destination buffers too small");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipMemcpyAsync(dst->d_nodes, ctx->bvh.nodes, nb, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dst->d_tri_pos, ctx->bvh.tri_pos, tb, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dst->d_tri_nor, ctx->bvh.tri_nor, tb, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

extern "C" int rt_bvh_adopt_device(rt_ctx* ctx, const rt_bvh_view* v) {
  if (!ctx) return RT_ERR_INVALID;
  if (!v || !v->d_nodes || !v->d_tri_pos || !v->d_tri_nor || v->n_nodes == 0 || v->n_tri == 0)
    return ctx->fail(RT_ERR_INVALID, "rt_bvh_adopt_device: incomplete view");
  if (v->nodes_bytes < 32ull * v->n_nodes || v->tri_pos_bytes < 48ull * v->n_tri || v->tri_nor_bytes < 48ull * v->n_tri)
    return ctx->fail(RT_ERR_INVALID, "rt_bvh_adopt_device: buffers smaller than the counts imply");
  { const int lrc = check_encoding_limits(ctx, "rt_bvh_adopt_device", v->n_nodes, v->n_tri); if (lrc != RT_OK) return lrc; }
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  ctx->have_bvh = false; ctx->built = false; ctx->tri_uv_n = 0;
  ctx->bvh = RtBvhDev();
  ctx->bvh.nodes = static_cast<float*>(v->d_nodes);
  ctx->bvh.tri_pos = static_cast<float*>(v->d_tri_pos);
  ctx->bvh.tri_nor = static_cast<float*>(v->d_tri_nor);
  ctx->bvh.n_nodes = v->n_nodes; ctx->bvh.n_tri = v->n_tri; ctx->bvh.owns_interchange = false;
  return finish_bvh(ctx);   // the pack kernel range-checks every child index and every leaf's [start, start + count) on the device
}

// ================================================================ material / texture shading (extension)
extern "C" int rt_set_surface(rt_ctx* ctx, const float* texcoords, uint32_t n_tex, const uint32_t* tex_idx, const int32_t* material_ids) {
  if (!ctx) return RT_ERR_INVALID;
  const uint32_t T = ctx->n_tri_mesh;
  if (T == 0) return ctx->fail(RT_ERR_INVALID, "rt_set_surface: call rt_set_mesh first");
  if (!texcoords || !tex_idx || n_tex == 0) return ctx->fail(RT_ERR_INVALID, "rt_set_surface: NULL / empty texcoords");
  for (size_t i = 0; i < (size_t)T * 3; ++i)   // a wild index would fault the GPU: check on the host
    if (tex_idx[i] >= n_tex) return ctx->fail(RT_ERR_INVALID, "rt_set_surface: texcoord index out of range");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  hipStream_t s = ctx->stream;
  HIPCHK(ctx->texcoords.ensure((size_t)n_tex * 2));
  HIPCHK(ctx->tidx.ensure((size_t)T * 3));
  HIPCHK(hipMemcpyAsync(ctx->texcoords.p, texcoords, sizeof(float) * 2 * n_tex, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(ctx->tidx.p, tex_idx, sizeof(uint32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, s));
  if (material_ids) {
    HIPCHK(ctx->matid.ensure(T));
    HIPCHK(hipMemcpyAsync(ctx->matid.p, material_ids, sizeof(int32_t) * (size_t)T, hipMemcpyHostToDevice, s));
  }
  ctx->surface_has_mat = material_ids != nullptr;
  ctx->surface_set = true;
  rt_launch_pack_mesh(ctx->pos.p, ctx->nor.p, ctx->vidx.p, ctx->nidx.p, ctx->texcoords.p, ctx->tidx.p, ctx->surface_has_mat ? ctx->matid.p : nullptr, T,
                      ctx->mesh_rec.p, nullptr, s);              // the records' texcoord / material slots
  HIPCHK(hipGetLastError());
  if (ctx->built && ctx->have_bvh && ctx->bvh.n_tri == T) {   // the LBVH of this mesh already exists: reorder now
    HIPCHK(ctx->tri_uv.ensure((size_t)T * 8));
    rt_launch_gather_uv(ctx->texcoords.p, ctx->tidx.p, ctx->surface_has_mat ? ctx->matid.p : nullptr, ctx->vals_a.p, T, ctx->tri_uv.p, s);
    HIPCHK(hipGetLastError());
    ctx->tri_uv_n = T;
  }
  HIPCHK(hipStreamSynchronize(s));
  return RT_OK;
}

extern "C" int rt_get_tri_uv(rt_ctx* ctx, float* tri_uv) {
  if (!ctx) return RT_ERR_INVALID;
  if (!tri_uv) return ctx->fail(RT_ERR_INVALID, "rt_get_tri_uv: NULL output");
  if (!ctx->have_bvh || ctx->tri_uv_n == 0 || ctx->tri_uv_n != ctx->bvh.n_tri)
    return ctx->fail(RT_ERR_INVALID, "rt_get_tri_uv: no surface records for the current BVH (rt_set_surface + rt_build_lbvh, or rt_set_tri_uv)");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(tri_uv, ctx->tri_uv.p, sizeof(float) * 8 * (size_t)ctx->tri_uv_n, hipMemcpyDeviceToHost));
  return RT_OK;
}

extern "C" int rt_set_tri_uv(rt_ctx* ctx, const float* tri_uv, uint32_t n_tri) {
  NEED_BVH();
  if (!tri_uv || n_tri != ctx->bvh.n_tri) return ctx->fail(RT_ERR_INVALID, "rt_set_tri_uv: need one record per triangle of the current BVH");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(ctx->tri_uv.ensure((size_t)n_tri * 8));
  HIPCHK(hipMemcpyAsync(ctx->tri_uv.p, tri_uv, sizeof(float) * 8 * (size_t)n_tri, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->tri_uv_n = n_tri;
  return RT_OK;
}

extern "C" int rt_tri_uv_device(rt_ctx* ctx, void** d_tri_uv, uint64_t* bytes) {
  NEED_BVH();
  if (!d_tri_uv || !bytes) return ctx->fail(RT_ERR_INVALID, "rt_tri_uv_device: NULL output");
  const bool have = ctx->tri_uv_n != 0 && ctx->tri_uv_n == ctx->bvh.n_tri;
  *d_tri_uv = have ? ctx->tri_uv.p : nullptr;
  *bytes = have ? 32ull * ctx->tri_uv_n : 0;
  return RT_OK;
}

extern "C" int rt_tri_uv_export_device(rt_ctx* ctx, void* d_dst, uint64_t dst_bytes) {
  NEED_BVH();
  if (ctx->tri_uv_n == 0 || ctx->tri_uv_n != ctx->bvh.n_tri) return ctx->fail(RT_ERR_INVALID, "rt_tri_uv_export_device: no surface records for the current BVH");
  if (!d_dst || dst_bytes < 32ull * ctx->tri_uv_n) return ctx->fail(RT_ERR_INVALID, "rt_tri_uv_export_device: destination too small");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipMemcpyAsync(d_dst, ctx->tri_uv.p, 32ull * ctx->tri_uv_n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

extern "C" int rt_adopt_tri_uv_device(rt_ctx* ctx, const void* d_tri_uv, uint32_t n_tri) {
  NEED_BVH();
  if (!d_tri_uv || n_tri != ctx->bvh.n_tri) return ctx->fail(RT_ERR_INVALID, "rt_adopt_tri_uv_device: need one record per triangle of the current BVH");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(ctx->tri_uv.ensure((size_t)n_tri * 8));
  HIPCHK(hipMemcpyAsync(ctx->tri_uv.p, d_tri_uv, 32ull * n_tri, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->tri_uv_n = n_tri;
  return RT_OK;
}

extern "C" int rt_set_materials(rt_ctx* ctx, const rt_material_rec* materials, uint32_t n_materials) {
  if (!ctx) return RT_ERR_INVALID;
  if (n_materials > 0 && !materials) return ctx->fail(RT_ERR_INVALID, "rt_set_materials: NULL materials");
  if (n_materials > (1u << 20)) return ctx->fail(RT_ERR_LIMIT, "rt_set_materials: more than 2^20 materials");
  for (uint32_t i = 0; i < n_materials; ++i)
    if (materials[i].texture >= RT_MAX_TEXTURES) return ctx->fail(RT_ERR_LIMIT, "rt_set_materials: texture slot >= RT_MAX_TEXTURES");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  std::vector<float> rec(4 * (size_t)std::max<uint32_t>(n_materials, 1), 0.0f);
  for (uint32_t i = 0; i < n_materials; ++i) {
    rec[4 * i + 0] = materials[i].kd[0]; rec[4 * i + 1] = materials[i].kd[1]; rec[4 * i + 2] = materials[i].kd[2];
    const uint32_t slot = materials[i].texture < 0 ? 0xFFFFFFFFu : (uint32_t)materials[i].texture;
    std::memcpy(&rec[4 * i + 3], &slot, 4);
  }
  HIPCHK(ctx->materials.ensure(std::max<uint32_t>(n_materials, 1)));
  HIPCHK(hipMemcpyAsync(ctx->materials.p, rec.data(), sizeof(float) * rec.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->materials_host.assign(materials, materials + n_materials);
  ctx->n_materials = n_materials;
  return RT_OK;
}

extern "C" int rt_set_texture(rt_ctx* ctx, uint32_t slot, const uint8_t* rgb8, uint32_t width, uint32_t height) {
  if (!ctx) return RT_ERR_INVALID;
  if (slot >= RT_MAX_TEXTURES) return ctx->fail(RT_ERR_LIMIT, "rt_set_texture: slot >= RT_MAX_TEXTURES");
  if (!rgb8 || width == 0 || height == 0 || width > 32768 || height > 32768) return ctx->fail(RT_ERR_INVALID, "rt_set_texture: bad image");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  const size_t n = (size_t)width * height;
  std::vector<uint32_t> px(n);
  for (size_t i = 0; i < n; ++i) px[i] = (uint32_t)rgb8[3 * i] | ((uint32_t)rgb8[3 * i + 1] << 8) | ((uint32_t)rgb8[3 * i + 2] << 16) | 0xFF000000u;
  HIPCHK(hipStreamSynchronize(ctx->stream));          // a frame in flight may still read the old texels
  HIPCHK(ctx->texels[slot].ensure(n));
  HIPCHK(hipMemcpy(ctx->texels[slot].p, px.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
  ctx->tex_host[slot].texels = ctx->texels[slot].p; ctx->tex_host[slot].w = width; ctx->tex_host[slot].h = height;
  ctx->n_textures = std::max(ctx->n_textures, slot + 1);
  HIPCHK(ctx->tex_table.ensure(RT_MAX_TEXTURES));
  HIPCHK(hipMemcpy(ctx->tex_table.p, ctx->tex_host, sizeof(ctx->tex_host), hipMemcpyHostToDevice));
  return RT_OK;
}

// ================================================================ hot path 2
extern "C" void rt_render_opts_default(rt_render_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->mode = RT_MODE_PT; o->gi_samples = 32; o->gi_bounces = 3; o->soft_samples = 4; o->max_stack = 12;   // compute.wgsl:25-27,259
  o->layout = RT_LAYOUT_WIDE64; o->stripe_rows = 0; o->stripe_rank = 0; o->stripe_world = 1;
}

// the constants each shader file is written with
extern "C" int rt_render_opts_for_mode(rt_render_opts* o, int mode) {
  if (!o || mode < RT_MODE_PRIMARY || mode > RT_MODE_FRAG_ROOT) return RT_ERR_INVALID;
  rt_render_opts_default(o);
  o->mode = mode;
  if (mode == RT_MODE_FRAG_WEB) { o->gi_samples = 32; o->gi_bounces = 3; o->soft_samples = 8; o->max_stack = 16; }    // web/shader.wgsl:22-24,273
  if (mode == RT_MODE_FRAG_ROOT) { o->gi_samples = 16; o->gi_bounces = 1; o->soft_samples = 8; o->max_stack = 16; }   // shader.wgsl:24-25,191,420
  return RT_OK;
}

extern "C" uint32_t rt_packed_rows(uint32_t height, int32_t stripe_rows, int32_t rank, int32_t world) {
  if (stripe_rows <= 0 || world <= 1) return height;
  const uint32_t sr = (uint32_t)stripe_rows;
  const uint32_t n_stripes = (height + sr - 1) / sr;
  uint32_t rows = 0;
  for (uint32_t s = (uint32_t)rank; s < n_stripes; s += (uint32_t)world) {
    const uint32_t y0 = s * sr;
    rows += (y0 + sr <= height) ? sr : (height - y0);
  }
  return rows;
}

namespace {
int check_opts(rt_ctx* ctx, const rt_render_opts* o) {
  if (!o) return ctx->fail(RT_ERR_INVALID, "render opts are NULL");
  if (o->mode < RT_MODE_PRIMARY || o->mode > RT_MODE_FRAG_ROOT) return ctx->fail(RT_ERR_INVALID, "bad mode");
  if (o->mode >= RT_MODE_FRAG_WEB) {
    if (o->soft_samples > 8) return ctx->fail(RT_ERR_INVALID, "RT_MODE_FRAG_*: soft_samples <= 8 (POISSON_DISK has 8 taps)");
    if (o->mode == RT_MODE_FRAG_ROOT && o->gi_samples > 16) return ctx->fail(RT_ERR_INVALID, "RT_MODE_FRAG_ROOT: gi_samples <= 16 (HEMI_SAMPLES has 16 directions)");
    if (o->layout != RT_LAYOUT_WIDE64) return ctx->fail(RT_ERR_INVALID, "RT_MODE_FRAG_*: only RT_LAYOUT_WIDE64");
    if (o->flags & RT_FLAG_TEXTURED) return ctx->fail(RT_ERR_INVALID, "RT_MODE_FRAG_*: the older shaders have a constant albedo; RT_FLAG_TEXTURED is not defined for them");
  }
  if (o->max_stack < 2 || o->max_stack > 32) return ctx->fail(RT_ERR_INVALID, "max_stack must be in [2,32]");
  if (o->gi_samples < 1 || o->gi_samples > 4096 || o->gi_bounces < 0 || o->gi_bounces > 64 || o->soft_samples < 1 || o->soft_samples > 64)
    return ctx->fail(RT_ERR_INVALID, "sample counts out of range");
  if (o->layout != RT_LAYOUT_WIDE64 && o->layout != RT_LAYOUT_AOS32) return ctx->fail(RT_ERR_INVALID, "bad layout");
  if (o->stripe_rows < 0 || (o->stripe_rows > 0 && (o->stripe_rows % 8 != 0)))
    return ctx->fail(RT_ERR_INVALID, "stripe_rows must be a multiple of 8 (tile height)");
  if (o->stripe_rows > 0 && (o->stripe_world < 1 || o->stripe_rank < 0 || o->stripe_rank >= o->stripe_world))
    return ctx->fail(RT_ERR_INVALID, "bad stripe rank/world");
  return RT_OK;
}
}  // namespace

namespace {
// What every frame pipeline takes from the context: the tree, the traversal layout, the step budget and (RT_FLAG_TEXTURED) the
// surface tables.  alloc_tables = false: the no-alloc entry point, which must find the material / texture tables in place.
int fill_scene_params(rt_ctx* ctx, const rt_render_opts* opts, RtRenderParams& p, bool alloc_tables) {
  p.nodes = ctx->bvh.nodes; p.tri_pos = ctx->bvh.tri_pos; p.tri_nor = ctx->bvh.tri_nor;
  p.wide = ctx->bvh.wide;
  if (ctx->have_paired) { const char* e = getenv("RTCORE_WIDE_PAIRED"); if (e && e[0] == '1') p.wide = ctx->wide_paired.p; }
  p.tri_pre = ctx->bvh.tri_pre; p.root_ref = ctx->bvh.root_ref; p.leaf_shift = ctx->bvh.leaf_shift;
  p.nodes_v2 = ctx->bvh.v2 ? 1 : 0;
  {   // (RTCORE_FAR_ADDRESSES=1: the 64-bit form everywhere - tests, A/B)
    const char* w = reinterpret_cast<const char*>(p.wide); const char* q = reinterpret_cast<const char*>(p.tri_pre);
    const char* lo = w < q ? w : q;
    const char* w_end = w + 64ull * ctx->bvh.n_nodes; const char* q_end = q + 48ull * ctx->bvh.n_tri;
    const char* hi = w_end > q_end ? w_end : q_end;
    const char* far = getenv("RTCORE_FAR_ADDRESSES");
    p.near32 = (w && q && (unsigned long long)(hi - lo) < 0xFFFFFF00ull && !(far && far[0] == '1')) ? 1u : 0u;
    p.near_base = lo; p.near_wide_off = (uint32_t)(w - lo); p.near_pre_off = (uint32_t)(q - lo);
  }
  // one budget step per loop iteration of the walk: WIDE64 spends one per internal record AND one per leaf triangle
  // (closest_wide advances leaf_j one triangle per iteration), AOS32 one per node visit with twice this budget
  p.step_budget = ctx->bvh.n_nodes + ctx->bvh.n_tri + 64u;
  if (!(opts->flags & RT_FLAG_TEXTURED)) return RT_OK;
  if (ctx->tri_uv_n == 0 || ctx->tri_uv_n != ctx->bvh.n_tri)
    return ctx->fail(RT_ERR_INVALID, "RT_FLAG_TEXTURED: no surface records for the current BVH (rt_set_surface before rt_build_lbvh, or rt_set_tri_uv)");
  for (uint32_t i = 0; i < ctx->n_materials; ++i) {
    const int32_t t = ctx->materials_host[i].texture;
    if (t >= 0 && ((uint32_t)t >= ctx->n_textures || ctx->tex_host[t].texels == nullptr))
      return ctx->fail(RT_ERR_INVALID, "RT_FLAG_TEXTURED: material " + std::to_string(i) + " names texture slot " + std::to_string(t) + ", which was never set");
  }
  if (!alloc_tables) {
    if (!ctx->materials.p || !ctx->tex_table.p)
      return ctx->fail(RT_ERR_INVALID, "rt_render_device_noalloc with RT_FLAG_TEXTURED: call rt_set_materials and rt_set_texture first (their tables are not allocated here)");
  } else {
    HIPCHK(ctx->materials.ensure(1)); HIPCHK(ctx->tex_table.ensure(RT_MAX_TEXTURES));
  }
  p.tri_uv = reinterpret_cast<const float4*>(ctx->tri_uv.p); p.materials = ctx->materials.p; p.textures = ctx->tex_table.p;
  p.n_materials = ctx->n_materials; p.n_textures = ctx->n_textures;
  return RT_OK;
}
}  // namespace

namespace {
// frame-coherent scheduling hints: one cost class per pixel of this image shape, cleared whenever the shape, the
// striping or the BVH changes (hints of another view are merely useless, never wrong: they only order the work)
// `moved`: the view differs from the last frame's (or, in a batch, between the batch's frames).
int setup_hints(rt_ctx* ctx, RtRenderParams& p, const rt_render_opts* opts, bool moved, int slot, hipStream_t s) {
  const uint64_t key = ((uint64_t)p.width << 40) ^ ((uint64_t)p.local_rows << 20) ^ ((uint64_t)p.stripe_rank << 8) ^ (uint64_t)p.stripe_world ^
                       ((uint64_t)ctx->bvh_serial << 52);
  const size_t n_px = (size_t)p.local_rows * p.width;
  if (key != ctx->pix_class_key || ctx->pix_class.n < n_px) {
    HIPCHK(hipStreamSynchronize(ctx->stream));   // frames in flight may still use the old array
    for (int k = 0; k < rt_ctx::MAX_FIF; ++k) if (ctx->fstream[k]) HIPCHK(hipStreamSynchronize(ctx->fstream[k]));
    HIPCHK(ctx->pix_class.ensure(n_px));
    HIPCHK(hipMemsetAsync(ctx->pix_class.p, 0, n_px, s));
    for (bool& r : ctx->class_partial_reset) r = true;
    ctx->pix_class_key = key;
  }
  if (ctx->class_partial_reset[slot]) {
    HIPCHK(hipMemsetAsync(p.class_hist, 0, sizeof(uint32_t) * 64, s));
    ctx->class_partial_reset[slot] = false;
  }
  p.pix_class = ctx->pix_class.p;
  // traversal steps per pixel: a ray that only crosses the root box takes a handful, a hit a few dozen
  // (cfg 3: ~6 / ~40, grazing rays up to 185).  A fixed t2 is scene-dependent (cfg 3 best at 40 steps: 0.523 -> 0.387 ms, 32: 0.47,
  // 48: 0.41, 64: 0.43; cfg 5 best at 80), so the hot threshold follows last frame's step histogram instead.
  // PT pixels are a chain of ~50 traversals whatever their class: only hit / miss is separated there.
  p.class_t1 = 16u;
  p.class_t2 = ((opts->mode == RT_MODE_PT && !p.pt_cap) || p.frag) ? 0xFFFFFFFFu : 0u;   // 0 = adaptive: the top class_hot_pct % of last frame's non-cheap pixels
  // A view that changed since the last frame moves the long rays by a few pixels: then a hot pixel promotes its 8x8
  // tile's non-cheap pixels and the hot share is smaller (tools/moving_camera.py, k_render with / without hints:
  // static view 0.397 vs 0.463 ms per-pixel, 0.414 promoted; model turning 0.25 deg per frame: per-pixel 0.462 = no
  // gain, promoted 0.438 = +6 %; 1 deg: +7 %; 4-15 deg: +2 %).  An unchanged view (what the reference redraws all the
  // time, script.js:865-974) keeps the sharper per-pixel classes: cfg 5 loses its whole gain when promoted (3.19 vs 3.51 ms).
  p.class_promote = moved ? 1u : 0u;
  if (const char* e = std::getenv("RTCORE_TILE_PROMOTE")) p.class_promote = (uint32_t)std::atoi(e);
  p.class_hot_pct = moved ? 10u : 20u;    // unpromoted, static view: 20 / 30 / 40 / 50 % -> k_render 0.394 / 0.415 / 0.394 / 0.451 ms (cfg 3), 3.20 / 3.21 / 3.23 / 3.27 ms (cfg 5)
  if (const char* e = std::getenv("RTCORE_HOT_PCT")) p.class_hot_pct = (uint32_t)std::atoll(e);
  {   // quantile bands of the non-cheap pixels, hottest last: with 5 classes the top 50 / 25 / 10 %
    const uint32_t n_hot = rt_n_class() - 2u;
    static const uint32_t shares[7][6] = {{0}, {20}, {35, 20}, {45, 25, 12}, {60, 40, 20, 8}, {65, 45, 30, 15, 6}, {70, 50, 35, 22, 12, 5}};
    for (uint32_t k = 0; k < 6; ++k) p.class_share[k] = k < n_hot ? shares[n_hot][k] : 0u;
    // Measured with 4 classes (k_render, one box, 3-class build beside it): cfg 3 0.388 vs 0.426 ms, cfg 5 3.08 vs 3.17,
    // S70k 4K 0.375 vs 0.381, 8-copy lattice 0.620 vs 0.625, S70k 1080p 0.222 vs 0.215 (the one loss).  The landscape is
    // jumpy: {40,15} is as good on cfg 3 but 12 % worse on S70k 1080p, {25,8} and {50,20} lose most of the gain.
    // PT part 1 keeps ONE hot class: the order of its records is also the order k_pt_samples walks them in, and finer
    // classes there cost 4 % (4.03 vs 3.86 ms).
    if (n_hot == 1 || std::getenv("RTCORE_HOT_PCT") || opts->mode == RT_MODE_PT) {
      p.class_share[0] = p.class_hot_pct;
      for (uint32_t k = 1; k < 6; ++k) p.class_share[k] = 0u;      // share 0: the class stays empty
    }
    else if (moved) for (uint32_t k = 0; k < n_hot; ++k) p.class_share[k] = (p.class_share[k] + 1u) / 2u;
    if (const char* e = std::getenv("RTCORE_CLASS_SHARES")) {   // e.g. "50,25,10"
      uint32_t k = 0;
      for (const char* q = e; *q && k < n_hot; ++k) { p.class_share[k] = (uint32_t)std::strtoul(q, const_cast<char**>(&q), 10); if (*q == ',') ++q; }
    }
  }
  if (const char* e = std::getenv("RTCORE_T1")) p.class_t1 = (uint32_t)std::atoll(e);
  if (const char* e = std::getenv("RTCORE_T2")) p.class_t2 = (uint32_t)std::atoll(e);
  return RT_OK;
}
}  // namespace

namespace {
// One frame.  noalloc = false: rt_render_device (scratch in lazily grown per-frame-slot allocations).  noalloc = true:
// rt_render_device_noalloc - the same kernels with this GPU's static scratch array carved up instead; no hipMalloc /
// hipFree anywhere on the path, hence no frame slots (the array is one per GPU), no scheduling hints, no PT split, no stats.
int render_frame(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts, void* d_rgba8, void* d_hit_tri,
                 void* d_hit_t, rt_render_stats* out, bool noalloc) {
  NEED_BVH();
  if (!uniforms) return ctx->fail(RT_ERR_INVALID, "uniforms are NULL");
  int rc = check_opts(ctx, opts);
  if (rc != RT_OK) return rc;
  if (((reinterpret_cast<uintptr_t>(d_rgba8) | reinterpret_cast<uintptr_t>(d_hit_tri) | reinterpret_cast<uintptr_t>(d_hit_t)) & 3u) != 0u)
    return ctx->fail(RT_ERR_INVALID, "output pointers must be 4-byte aligned");
  const float fw = uniforms[36], fh = uniforms[37];
  if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw > 65536.0f || fh > 65536.0f) return ctx->fail(RT_ERR_INVALID, "resolution out of range");
  DeviceGuard g(ctx->device);
  RtRenderParams p;
  std::memset(&p, 0, sizeof(p));
  rc = fill_scene_params(ctx, opts, p, /*alloc_tables=*/!noalloc);
  if (rc != RT_OK) return rc;
  std::memcpy(p.U, uniforms, sizeof(float) * 44);
  p.mode = opts->mode; p.gi_samples = opts->gi_samples; p.gi_bounces = opts->gi_bounces; p.soft_samples = opts->soft_samples;
  p.max_stack = opts->max_stack;
  p.frag = opts->mode == RT_MODE_FRAG_WEB ? 1 : (opts->mode == RT_MODE_FRAG_ROOT ? 2 : 0);
  p.miss_rgba = p.frag ? 0xFFFFCC99u : 0xFF000000u;   // rgba8unorm of (0.6, 0.8, 1, 1) (web/shader.wgsl:391) / (0, 0, 0, 1) (compute.wgsl:366)
  p.width = (uint32_t)fw; p.height = (uint32_t)fh;
  const bool striped = opts->stripe_rows > 0 && opts->stripe_world > 1;
  p.stripe_rows = striped ? opts->stripe_rows : 0; p.stripe_rank = striped ? opts->stripe_rank : 0;
  p.stripe_world = striped ? opts->stripe_world : 1;
  p.local_rows = rt_packed_rows(p.height, p.stripe_rows, p.stripe_rank, p.stripe_world);
  p.tiles_x = (p.width + 7) / 8;
  p.n_tiles = p.tiles_x * ((p.local_rows + 7) / 8);
  p.rgba8 = static_cast<uint8_t*>(d_rgba8); p.hit_tri = static_cast<uint32_t*>(d_hit_tri); p.hit_t = static_cast<float*>(d_hit_t);
  const uint32_t cull_blocks = rt_cull_blocks(p.n_tiles);
  // Frame slot + stream.  Synchronous frames (stats wanted) and contexts with one frame in flight use the context
  // stream and slot 0 after joining whatever is still running; asynchronous frames with two in flight alternate
  // between the two frame streams, each behind everything queued on the context stream so far.
  const bool overlap = !noalloc && ctx->frames_in_flight >= 2 && out == nullptr;
  int slot = 0;
  hipStream_t s = ctx->stream;
  SlotTicket ticket(ctx);
  if (overlap) {
    slot = (int)(ctx->f_count % (uint32_t)ctx->frames_in_flight);
    ticket.take(slot);
    s = ctx->fstream[slot];
    HIPCHK(hipEventRecord(ctx->f_main, ctx->stream));
    HIPCHK(hipStreamWaitEvent(s, ctx->f_main, 0));
  } else {
    JOIN_FRAMES();
  }
  // RT_MODE_PRIMARY / SHADOW frames run as two launches (rt_trace.hip KIND_TRAVERSE / KIND_SHADE); the no-alloc entry point keeps
  // the whole pixel in one launch (its static array has no room for a hit buffer), RTCORE_TWO_PHASE=0 does so everywhere (A/B)
  // A SINGLE frame keeps the whole pixel in one launch: its launch time is its longest chain of dependent fetches, which more
  // resident waves only stretch (k_render 0.54 ms against 0.49 + 0.19 ms for the two launches, profiles/r3/ab_two_phase.txt);
  // RTCORE_TWO_PHASE=1 forces the split (A/B, tests)
  bool two_phase = false;
  if (const char* e = std::getenv("RTCORE_TWO_PHASE")) two_phase = !noalloc && (opts->mode == RT_MODE_PRIMARY || opts->mode == RT_MODE_SHADOW) && std::atoi(e) != 0;
  const size_t tc_words = 3 * (size_t)rt_shard_words() + 32 + 64 + (size_t)rt_n_class() * cull_blocks;
  const size_t st_words = 8 + 32 + (size_t)rt_render_grid(ctx->n_cu) * 8 + (size_t)rt_n_class() * p.n_tiles;
  const size_t list_words = (size_t)p.local_rows * p.width;
  uint32_t* sc_tc; unsigned long long* sc_st; uint32_t* sc_list;
  if (noalloc) {
    if (!ctx->noalloc_scratch) return ctx->fail(RT_ERR_INVALID, "rt_render_device_noalloc: no static scratch on this device (device id >= 64)");
    if (list_words > NOALLOC_MAX_PIXELS || p.n_tiles > NOALLOC_MAX_TILES || rt_render_grid(ctx->n_cu) > NOALLOC_MAX_GRID ||
        rt_n_class() != (uint32_t)RT_N_CLASS || list_words * sizeof(uint32_t) > NOALLOC_LIST_BYTES ||
        st_words * sizeof(unsigned long long) > NOALLOC_STATS_BYTES || tc_words * sizeof(uint32_t) > NOALLOC_TC_BYTES)
      return ctx->fail(RT_ERR_LIMIT, "rt_render_device_noalloc: the frame (or this rank's stripe share) has " + std::to_string(list_words) +
                                     " pixels / " + std::to_string(p.n_tiles) + " 8x8 tiles; the static scratch holds at most " +
                                     std::to_string(NOALLOC_MAX_PIXELS) + " pixels (3840x2160) / " + std::to_string(NOALLOC_MAX_TILES) +
                                     " tiles - use rt_render_device");
    sc_list = reinterpret_cast<uint32_t*>(ctx->noalloc_scratch);
    sc_st = reinterpret_cast<unsigned long long*>(ctx->noalloc_scratch + NOALLOC_OFF_STATS);
    sc_tc = reinterpret_cast<uint32_t*>(ctx->noalloc_scratch + NOALLOC_OFF_TC);
  } else {
    HIPCHK(ctx->r_tile_counter[slot].ensure(tc_words));
    const unsigned long long* st_before = ctx->r_stats[slot].p;
    HIPCHK(ctx->r_stats[slot].ensure(st_words));
    if (ctx->r_stats[slot].p != st_before) ctx->class_partial_reset[slot] = true;   // the step histogram lives there
    HIPCHK(ctx->r_list[slot].ensure(list_words));
    if (two_phase) HIPCHK(ctx->r_hits[slot].ensure(list_words));
    sc_tc = ctx->r_tile_counter[slot].p; sc_st = ctx->r_stats[slot].p; sc_list = ctx->r_list[slot].p;
  }
  p.shard_stride = rt_shard_stride();
  p.blocks_per_cu_cap = overlap ? 2u : 0u;
  p.shard_group = 8;
  // PT split (sample-parallel GI) wherever its fixed-size term registers suffice; RTCORE_PT_SPLIT=0 keeps one lane per pixel
  {
    bool split = !noalloc && opts->mode == RT_MODE_PT && opts->gi_samples >= 2 && opts->gi_bounces >= 1 && opts->gi_bounces <= 4;
    if (const char* e = std::getenv("RTCORE_PT_SPLIT")) split = split && std::atoi(e) != 0;
    p.pt_cap = 0;
    if (split && p.n_tiles > 0) p.pt_cap = 1;   // sized below, once shard_group is final
  }
  if (const char* e = std::getenv("RTCORE_SHARD_GROUP")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) p.shard_group = (uint32_t)v; }
  if (p.pt_cap) {
    p.pt_cap = rt_pt_capacity(p.local_rows * p.width, p.shard_group);
    const size_t n_terms = (size_t)64 * p.pt_cap * (size_t)opts->gi_samples * ((size_t)opts->gi_bounces * 3 + 1);
    if (n_terms * sizeof(float) > ((size_t)12 << 30)) p.pt_cap = 0;    // the bounce terms would not fit a modest budget: one lane per pixel
    else {
      HIPCHK(ctx->pt_recs[slot].ensure((size_t)64 * p.pt_cap * 3));
      HIPCHK(ctx->pt_terms[slot].ensure(n_terms));
      p.pt_recs = ctx->pt_recs[slot].p; p.pt_terms = ctx->pt_terms[slot].p;
    }
  }
  // Everything k_render reads is (re)written by this frame's own k_cull / k_compact first: the queue heads are zeroed by
  // k_cull, block totals / masks of every tile / the list and its counts are written in full.  (Nothing else has to start
  // at zero, which is what lets the no-alloc variant reuse one array call after call.)
  p.shard_heads = sc_tc; p.list_count = sc_tc + rt_shard_words();
  p.pt_heads = sc_tc + rt_shard_words() + 32;
  p.pt_count = p.pt_heads + rt_shard_words();
  p.block_total = p.pt_count + 64;
  if (two_phase) { p.shade_heads = p.block_total + (size_t)rt_n_class() * cull_blocks; p.hits = ctx->r_hits[slot].p; }
  p.list = sc_list;
  p.stats = sc_st; p.stats_partial = sc_st + 8 + 32;
  p.class_hist = reinterpret_cast<uint32_t*>(sc_st + 8);   // 64 u32
  p.masks = sc_st + 8 + 32 + (size_t)rt_render_grid(ctx->n_cu) * 8;
  if (ctx->coherent_scheduling && !noalloc) {
    const bool moved = std::memcmp(ctx->last_uniforms, uniforms, sizeof(float) * 44) != 0;
    std::memcpy(ctx->last_uniforms, uniforms, sizeof(float) * 44);
    rc = setup_hints(ctx, p, opts, moved, slot, s);
    if (rc != RT_OK) return rc;
  }
  rt_cull_rect(p, ctx->bvh.root_box);
  if (p.frag) {   // the fragment shaders' screen y runs the other way (uv.y = 1 - (gy + 0.5) / h): the same rectangle, rows mirrored
    const int y0 = (int)p.height - 1 - p.cull_y1, y1 = (int)p.height - 1 - p.cull_y0;
    p.cull_y0 = y0; p.cull_y1 = y1;
  }
  if (out && p.n_tiles == 0) HIPCHK(hipMemsetAsync(p.stats, 0, sizeof(unsigned long long) * 8, s));
  hipEvent_t* E = nullptr;   // {frame begin, k_render begin, k_render end, frame end}
  if (ctx->prof_on) { E = &ctx->prof_ev[5 * (size_t)(ctx->prof_n % rt_ctx::PROF_RING)]; ctx->prof_n += (p.n_tiles > 0); }
  else if (out) E = ctx->ev;
  if (p.n_tiles > 0) {
    if (E) HIPCHK(hipEventRecord(E[0], s));
    rt_launch_render(p, opts->layout, (opts->flags & RT_FLAG_COUNT_WORK) != 0, out != nullptr, ctx->n_cu, s, E ? E[1] : nullptr, E ? E[2] : nullptr, E ? E[4] : nullptr);
    HIPCHK(hipGetLastError());
    if (E) HIPCHK(hipEventRecord(E[3], s));
  }
  if (overlap) {
    HIPCHK(hipEventRecord(ctx->f_done[slot], s));
    ctx->f_pending[slot] = true;
    ticket.done = true;
  }
  if (out) {
    unsigned long long st[8];
    uint32_t traced = 0;
    HIPCHK(hipMemcpyAsync(st, p.stats, sizeof(st), hipMemcpyDeviceToHost, s));
    if (p.n_tiles > 0) HIPCHK(hipMemcpyAsync(&traced, p.list_count, sizeof(traced), hipMemcpyDeviceToHost, s));   // k_compact: entries of the pixel list
    HIPCHK(hipStreamSynchronize(s));
    std::memset(out, 0, sizeof(*out));
    out->pixels_traced = traced;
    out->rays_primary = st[0]; out->rays_shadow = st[1]; out->rays_bounce = st[2];
    out->box_tests = st[3]; out->tri_tests = st[4]; out->hit_updates = st[5]; out->pixels = st[6]; out->stack_drops = st[7];
    if (p.n_tiles > 0) {
      (void)hipEventElapsedTime(&out->ms_kernel, E[0], E[3]);
      (void)hipEventElapsedTime(&out->ms_traverse, E[1], E[2]);
    }
  }
  return RT_OK;
}
}  // namespace

namespace {
// A batch of frames of one image shape (rt_render_batch_device): the frames' k_cull / k_compact passes run as the y
// dimension of two launches and ONE persistent k_render drains all their pixel lists, hottest classes of every frame
// first - the tail of one frame is filled by the next frame's pixels instead of idling the chip, and the four launches'
// fixed cost is paid once.  Same per-pixel code as render_frame, hence the same bytes.
int render_batch(rt_ctx* ctx, uint32_t n_frames, const float* uniforms, const rt_render_opts* opts, void* const* d_rgba8,
                 void* const* d_hit_tri, void* const* d_hit_t) {
  NEED_BVH();
  if (n_frames < 1 || n_frames > (uint32_t)RT_MAX_BATCH) return ctx->fail(RT_ERR_INVALID, "rt_render_batch_device: n_frames must be in [1, " + std::to_string(RT_MAX_BATCH) + "]");
  if (!uniforms) return ctx->fail(RT_ERR_INVALID, "uniforms are NULL");
  int rc = check_opts(ctx, opts);
  if (rc != RT_OK) return rc;
  const int frag = opts->mode == RT_MODE_FRAG_WEB ? 1 : (opts->mode == RT_MODE_FRAG_ROOT ? 2 : 0);
  if (opts->mode > RT_MODE_PT && !frag) return ctx->fail(RT_ERR_INVALID, "rt_render_batch_device: unknown mode");
  if (frag && ((opts->flags & RT_FLAG_TEXTURED) || opts->layout != RT_LAYOUT_WIDE64))
    return ctx->fail(RT_ERR_INVALID, "rt_render_batch_device: the fragment-shader modes take WIDE64 records and no RT_FLAG_TEXTURED (as rt_render_device)");
  const float fw = uniforms[36], fh = uniforms[37];
  if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw > 65536.0f || fh > 65536.0f) return ctx->fail(RT_ERR_INVALID, "resolution out of range");
  RtBatch bt;
  std::memset(&bt, 0, sizeof(bt));
  bt.n = n_frames;
  RtFrameRec recs[RT_MAX_BATCH];
  std::memset(recs, 0, sizeof(RtFrameRec) * n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) {
    const float* U = uniforms + 44 * (size_t)f;
    if (U[36] != fw || U[37] != fh) return ctx->fail(RT_ERR_INVALID, "rt_render_batch_device: every frame of a batch has the same resolution (frame " + std::to_string(f) + " differs)");
    void* o[3] = {d_rgba8 ? d_rgba8[f] : nullptr, d_hit_tri ? d_hit_tri[f] : nullptr, d_hit_t ? d_hit_t[f] : nullptr};
    if (((reinterpret_cast<uintptr_t>(o[0]) | reinterpret_cast<uintptr_t>(o[1]) | reinterpret_cast<uintptr_t>(o[2])) & 3u) != 0u)
      return ctx->fail(RT_ERR_INVALID, "output pointers must be 4-byte aligned");
    std::memcpy(recs[f].U, U, sizeof(float) * RT_U_FLOATS);
    recs[f].rgba8 = static_cast<uint8_t*>(o[0]); recs[f].hit_tri = static_cast<uint32_t*>(o[1]); recs[f].hit_t = static_cast<float*>(o[2]);
    // frames whose 44 uniforms are byte-identical to an earlier frame's share that frame's cull masks and pixel list: one
    // k_scan / k_compact per distinct view (every frame is still traced and written in full)
    recs[f].src = f;
    for (uint32_t g = 0; g < f; ++g)
      if (std::memcmp(uniforms + 44 * (size_t)g, U, sizeof(float) * 44) == 0) { recs[f].src = g; break; }
  }
  DeviceGuard g(ctx->device);
  RtRenderParams p;
  std::memset(&p, 0, sizeof(p));
  rc = fill_scene_params(ctx, opts, p, /*alloc_tables=*/true);
  if (rc != RT_OK) return rc;
  std::memcpy(p.U, uniforms, sizeof(float) * 44);               // (the kernels of a batch read bt.frames[f].U)
  p.mode = opts->mode; p.gi_samples = opts->gi_samples; p.gi_bounces = opts->gi_bounces; p.soft_samples = opts->soft_samples;
  p.max_stack = opts->max_stack;
  p.frag = frag;
  p.miss_rgba = frag ? 0xFFFFCC99u : 0xFF000000u;               // as render_frame
  p.width = (uint32_t)fw; p.height = (uint32_t)fh;
  const bool striped = opts->stripe_rows > 0 && opts->stripe_world > 1;
  p.stripe_rows = striped ? opts->stripe_rows : 0; p.stripe_rank = striped ? opts->stripe_rank : 0;
  p.stripe_world = striped ? opts->stripe_world : 1;
  p.local_rows = rt_packed_rows(p.height, p.stripe_rows, p.stripe_rank, p.stripe_world);
  p.tiles_x = (p.width + 7) / 8;
  p.n_tiles = p.tiles_x * ((p.local_rows + 7) / 8);
  if (p.n_tiles == 0) return RT_OK;                             // this rank owns no rows
  const uint32_t cull_blocks = rt_cull_blocks(p.n_tiles);
  // Pipelines in flight, as render_frame: with rt_set_frames_in_flight(n >= 2) batches rotate over the n frame streams (own
  // scratch each), so the cull / compaction passes and the head of batch i+1 run beside the tail of batch i.
  const bool overlap = ctx->frames_in_flight >= 2;
  const size_t list_words = ((size_t)p.local_rows * p.width + 63u) & ~(size_t)63u;
  if (list_words * n_frames > 0xFFFFFFFFull) return ctx->fail(RT_ERR_LIMIT, "rt_render_batch_device: the batch's pixel lists exceed 2^32 entries");
  int slot = 0;
  hipStream_t s = ctx->stream;
  SlotTicket ticket(ctx);
  if (overlap) {
    slot = (int)(ctx->f_count % (uint32_t)ctx->frames_in_flight);
    ticket.take(slot);
    s = ctx->fstream[slot];
    HIPCHK(hipEventRecord(ctx->f_main, ctx->stream));
    HIPCHK(hipStreamWaitEvent(s, ctx->f_main, 0));
  } else {
    JOIN_FRAMES();
  }
  p.blocks_per_cu_cap = overlap ? 2u : 0u;
  bt.mask_stride = rt_n_class() * p.n_tiles;
  bt.bt_stride = rt_n_class() * cull_blocks;
  bt.list_stride = (uint32_t)list_words;
  bt.lc_stride = RT_LC_WORDS;
  // The traverse / shade split (two launches) is an option, not the default: what it bought was occupancy, and the whole-pixel
  // kernel capped at 96 VGPRs reaches the same 5 waves per SIMD in ONE launch - as fast on whole frames (0.2339 / 0.2350 against
  // 0.2385 / 0.2373 ms per frame) and 8 - 11 % faster on stripe shares, where the second launch is a second tail
  // (profiles/r3/ab_register_cap.txt, project_scaling_*.txt).  RTCORE_TWO_PHASE=1 selects the split.
  bool two_phase = false;
  if (const char* e = std::getenv("RTCORE_TWO_PHASE")) two_phase = std::atoi(e) != 0 && (opts->mode == RT_MODE_PRIMARY || opts->mode == RT_MODE_SHADOW);
  const size_t tc_words = 2 * (size_t)rt_shard_words() + (size_t)RT_LC_WORDS * n_frames + (size_t)bt.bt_stride * n_frames;
  const size_t st_words = 8 + 32 + (size_t)bt.mask_stride * n_frames;
  HIPCHK(ctx->r_tile_counter[slot].ensure(tc_words));
  const unsigned long long* st_before = ctx->r_stats[slot].p;
  HIPCHK(ctx->r_stats[slot].ensure(st_words));
  if (ctx->r_stats[slot].p != st_before) ctx->class_partial_reset[slot] = true;
  HIPCHK(ctx->r_list[slot].ensure(list_words * n_frames));
  if (two_phase) HIPCHK(ctx->r_hits[slot].ensure(list_words * n_frames));
  const bool count_work = (opts->flags & RT_FLAG_COUNT_WORK) != 0;
  if (count_work) HIPCHK(ctx->r_bstats[slot].ensure((size_t)RT_MAX_BATCH * 9 + (size_t)n_frames * rt_render_grid(ctx->n_cu) * 4 * 8));
  uint32_t* sc_tc = ctx->r_tile_counter[slot].p;
  unsigned long long* sc_st = ctx->r_stats[slot].p;
  p.shard_stride = rt_shard_stride();
  p.shard_group = 8;
  if (const char* e = std::getenv("RTCORE_SHARD_GROUP")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) p.shard_group = (uint32_t)v; }
  p.shard_heads = sc_tc; p.list_count = sc_tc + rt_shard_words();
  p.block_total = p.list_count + (size_t)RT_LC_WORDS * n_frames;
  if (two_phase) { p.shade_heads = p.block_total + (size_t)bt.bt_stride * n_frames; p.hits = ctx->r_hits[slot].p; }
  p.list = ctx->r_list[slot].p;
  p.stats = sc_st;
  p.class_hist = reinterpret_cast<uint32_t*>(sc_st + 8);        // where render_frame keeps it: single frames and batches share the history
  p.masks = sc_st + 8 + 32;
  if (count_work) { p.stats = ctx->r_bstats[slot].p; p.stats_partial = ctx->r_bstats[slot].p + (size_t)RT_MAX_BATCH * 9; }
  if (ctx->coherent_scheduling) {
    bool moved = std::memcmp(ctx->last_uniforms, uniforms, sizeof(float) * 44) != 0;
    for (uint32_t f = 1; f < n_frames; ++f) moved = moved || std::memcmp(uniforms, uniforms + 44 * (size_t)f, sizeof(float) * 44) != 0;
    std::memcpy(ctx->last_uniforms, uniforms + 44 * (size_t)(n_frames - 1), sizeof(float) * 44);
    // Hints order a batch only while the view stands still.  Under motion the newest hints are up to a batch old (8
    // frames at 1 degree each) and the batch's own length already hides the tail they are for: tools/batch_gain.py,
    // 8 frames, turning 1 degree per frame, k_render roofline fraction with / without hints: cfg 3 0.883 / 0.932, S70k 1080p
    // 0.717 / 0.764, a 1/8 stripe share 0.596 / 0.581; static view 0.964 / 0.948, 0.778 / 0.758, 0.701 / 0.604.
    if (!moved) {
      rc = setup_hints(ctx, p, opts, false, slot, s);
      if (rc != RT_OK) return rc;
    }
  }
  for (uint32_t f = 0; f < n_frames; ++f) {
    RtRenderParams pf = p;
    std::memcpy(pf.U, uniforms + 44 * (size_t)f, sizeof(float) * 44);
    rt_cull_rect(pf, ctx->bvh.root_box);
    if (frag) { const int y0 = (int)p.height - 1 - pf.cull_y1, y1 = (int)p.height - 1 - pf.cull_y0; pf.cull_y0 = y0; pf.cull_y1 = y1; }   // rows mirrored, as render_frame
    recs[f].cull[0] = pf.cull_x0; recs[f].cull[1] = pf.cull_x1; recs[f].cull[2] = pf.cull_y0; recs[f].cull[3] = pf.cull_y1;
  }
  HIPCHK(ctx->r_frames[slot].ensure(RT_MAX_BATCH));
  rt_launch_put_frames(recs, n_frames, ctx->r_frames[slot].p, s);     // (the slot's previous batch ran on this same stream: in order)
  bt.frames = ctx->r_frames[slot].p;
  hipEvent_t* E = nullptr;   // {batch begin, k_render begin, k_render end, batch end}: rt_profile_end reports per batch
  if (ctx->prof_on) { E = &ctx->prof_ev[5 * (size_t)(ctx->prof_n % rt_ctx::PROF_RING)]; ctx->prof_n += 1; }
  if (E) HIPCHK(hipEventRecord(E[0], s));
  rt_launch_render_batch(p, bt, opts->layout, count_work, ctx->n_cu, s, E ? E[1] : nullptr, E ? E[2] : nullptr, E ? E[4] : nullptr);
  HIPCHK(hipGetLastError());
  if (count_work) { ctx->bstats_slot = slot; ctx->bstats_n = n_frames; }
  if (E) HIPCHK(hipEventRecord(E[3], s));
  if (overlap) {
    HIPCHK(hipEventRecord(ctx->f_done[slot], s));
    ctx->f_pending[slot] = true;
    ticket.done = true;
  }
  return RT_OK;
}
}  // namespace

extern "C" int rt_render_device(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts, void* d_rgba8, void* d_hit_tri,
                                void* d_hit_t, rt_render_stats* out) {
  return render_frame(ctx, uniforms, opts, d_rgba8, d_hit_tri, d_hit_t, out, /*noalloc=*/false);
}

// No hipMalloc / hipFree / allocating routine on this path: scratch = this GPU's static array (see the top of this file).
extern "C" int rt_render_device_noalloc(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts, void* d_rgba8,
                                        void* d_hit_tri, void* d_hit_t) {
  return render_frame(ctx, uniforms, opts, d_rgba8, d_hit_tri, d_hit_t, nullptr, /*noalloc=*/true);
}

extern "C" int rt_render_batch_device(rt_ctx* ctx, uint32_t n_frames, const float* uniforms, const rt_render_opts* opts,
                                      void* const* d_rgba8, void* const* d_hit_tri, void* const* d_hit_t) {
  if (!ctx) return RT_ERR_INVALID;
  return render_batch(ctx, n_frames, uniforms, opts, d_rgba8, d_hit_tri, d_hit_t);
}

// Counters of the last batch that was rendered with RT_FLAG_COUNT_WORK, one rt_render_stats per frame (synchronises).
extern "C" int rt_batch_stats(rt_ctx* ctx, uint32_t n_frames, rt_render_stats* out) {
  if (!ctx) return RT_ERR_INVALID;
  if (!out) return ctx->fail(RT_ERR_INVALID, "rt_batch_stats: NULL output");
  if (ctx->bstats_slot < 0 || ctx->bstats_n == 0) return ctx->fail(RT_ERR_INVALID, "rt_batch_stats: no batch has been rendered with RT_FLAG_COUNT_WORK");
  if (n_frames != ctx->bstats_n) return ctx->fail(RT_ERR_INVALID, "rt_batch_stats: the last counted batch had " + std::to_string(ctx->bstats_n) + " frames");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  unsigned long long st[RT_MAX_BATCH * 9];
  HIPCHK(hipMemcpy(st, ctx->r_bstats[ctx->bstats_slot].p, sizeof(unsigned long long) * 9 * n_frames, hipMemcpyDeviceToHost));
  for (uint32_t f = 0; f < n_frames; ++f) {
    const unsigned long long* v = st + 9 * (size_t)f;
    rt_render_stats& o = out[f];
    std::memset(&o, 0, sizeof(o));
    o.rays_primary = v[0]; o.rays_shadow = v[1]; o.rays_bounce = v[2]; o.box_tests = v[3]; o.tri_tests = v[4]; o.hit_updates = v[5];
    o.pixels = v[6]; o.stack_drops = v[7]; o.pixels_traced = v[8];
  }
  return RT_OK;
}

extern "C" int rt_set_coherent_scheduling(rt_ctx* ctx, int on) {
  if (!ctx) return RT_ERR_INVALID;
  ctx->coherent_scheduling = on != 0;
  ctx->pix_class_key = ~0ull;        // hints start from scratch when switched back on
  return RT_OK;
}

extern "C" int rt_set_frames_in_flight(rt_ctx* ctx, int n) {
  if (!ctx) return RT_ERR_INVALID;
  if (n < 1 || n > rt_ctx::MAX_FIF) return ctx->fail(RT_ERR_INVALID, "rt_set_frames_in_flight: 1 .. 4");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n && n >= 2; ++k)
    if (!ctx->fstream[k]) {
      HIPCHK(hipStreamCreateWithFlags(&ctx->fstream[k], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ctx->f_done[k], hipEventDisableTiming));
    }
  if (n >= 2 && !ctx->f_main) HIPCHK(hipEventCreateWithFlags(&ctx->f_main, hipEventDisableTiming));
  ctx->frames_in_flight = n;
  return RT_OK;
}

extern "C" int rt_frames_join(rt_ctx* ctx, int keep_latest) {
  if (!ctx) return RT_ERR_INVALID;
  if (keep_latest < 0 || keep_latest >= rt_ctx::MAX_FIF) return ctx->fail(RT_ERR_INVALID, "rt_frames_join: keep_latest is 0 .. 3");
  DeviceGuard g(ctx->device);
  for (int k = 0; k < rt_ctx::MAX_FIF; ++k) {
    // the keep_latest most recently issued frames stay un-joined: issue numbers f_count - keep_latest .. f_count - 1
    if (!ctx->f_pending[k] || ctx->f_count - ctx->f_seq[k] <= (uint32_t)keep_latest) continue;
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->f_done[k], 0));
    ctx->f_pending[k] = false;
  }
  return RT_OK;
}

extern "C" int rt_profile_begin(rt_ctx* ctx) {
  if (!ctx) return RT_ERR_INVALID;
  DeviceGuard g(ctx->device);
  if (ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(5 * (size_t)rt_ctx::PROF_RING, nullptr);
    for (auto& e : ctx->prof_ev) HIPCHK(hipEventCreate(&e));
  }
  ctx->prof_on = true; ctx->prof_n = 0;
  return RT_OK;
}

extern "C" int rt_profile_end_phases(rt_ctx* ctx, float* ms_frame_avg, float* ms_traverse_avg, float* ms_shade_avg, uint32_t* n_frames) {
  if (!ctx) return RT_ERR_INVALID;
  if (!ctx->prof_on) return ctx->fail(RT_ERR_INVALID, "rt_profile_end without rt_profile_begin");
  DeviceGuard g(ctx->device);
  JOIN_FRAMES();
  ctx->prof_on = false;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint32_t n = std::min<uint32_t>(ctx->prof_n, rt_ctx::PROF_RING);   // the ring keeps the last PROF_RING frames
  double fsum = 0.0, tsum = 0.0, ssum = 0.0;
  for (uint32_t i = 0; i < n; ++i) {   // {pipeline begin, traversal begin, last traversal launch's end, pipeline end, first traversal launch's end}
    const hipEvent_t* E = &ctx->prof_ev[5 * (size_t)i];
    float a = 0.f, b = 0.f, c = 0.f;
    HIPCHK(hipEventElapsedTime(&a, E[0], E[3]));
    HIPCHK(hipEventElapsedTime(&b, E[1], E[4]));
    HIPCHK(hipEventElapsedTime(&c, E[4], E[2]));
    if (c < 0.02f) { b += c; c = 0.f; }            // nothing was launched between the two events (no second phase): two records 2 - 5 us apart
    fsum += a; tsum += b; ssum += c;
  }
  if (ms_frame_avg) *ms_frame_avg = n ? (float)(fsum / n) : 0.f;
  if (ms_traverse_avg) *ms_traverse_avg = n ? (float)(tsum / n) : 0.f;
  if (ms_shade_avg) *ms_shade_avg = n ? (float)(ssum / n) : 0.f;
  if (n_frames) *n_frames = n;
  return RT_OK;
}

extern "C" int rt_profile_end(rt_ctx* ctx, float* ms_frame_avg, float* ms_traverse_avg, uint32_t* n_frames) {
  float t = 0.f, sh = 0.f;
  const int rc = rt_profile_end_phases(ctx, ms_frame_avg, &t, &sh, n_frames);
  if (rc == RT_OK && ms_traverse_avg) *ms_traverse_avg = t + sh;
  return rc;
}

extern "C" int rt_render(rt_ctx* ctx, const float uniforms[44], const rt_render_opts* opts, uint8_t* rgba8, uint32_t* hit_tri,
                         float* hit_t, rt_render_stats* out) {
  NEED_BVH();
  if (!uniforms) return ctx->fail(RT_ERR_INVALID, "uniforms are NULL");
  int rc = check_opts(ctx, opts);
  if (rc != RT_OK) return rc;
  const float fw = uniforms[36], fh = uniforms[37];
  if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw > 65536.0f || fh > 65536.0f) return ctx->fail(RT_ERR_INVALID, "resolution out of range");
  const uint32_t w = (uint32_t)fw, h = (uint32_t)fh;
  const bool striped = opts->stripe_rows > 0 && opts->stripe_world > 1;
  const uint32_t rows = rt_packed_rows(h, striped ? opts->stripe_rows : 0, opts->stripe_rank, striped ? opts->stripe_world : 1);
  const size_t n = (size_t)rows * w;
  DeviceGuard g(ctx->device);
  if (rgba8) HIPCHK(ctx->o_rgba.ensure(n * 4));
  if (hit_tri) HIPCHK(ctx->o_tri.ensure(n));
  if (hit_t) HIPCHK(ctx->o_t.ensure(n));
  rt_render_stats st;
  rc = rt_render_device(ctx, uniforms, opts, rgba8 ? ctx->o_rgba.p : nullptr, hit_tri ? ctx->o_tri.p : nullptr,
                        hit_t ? ctx->o_t.p : nullptr, &st);
  if (rc != RT_OK) return rc;
  if (out) *out = st;
  auto fetch = [&](void* host, const void* dev, size_t elem) -> int {
    if (!host || n == 0) return RT_OK;
    if (!striped) { HIPCHK(hipMemcpy(host, dev, n * elem, hipMemcpyDeviceToHost)); return RT_OK; }
    std::vector<unsigned char> tmp(n * elem);
    HIPCHK(hipMemcpy(tmp.data(), dev, n * elem, hipMemcpyDeviceToHost));
    const uint32_t sr = (uint32_t)opts->stripe_rows;
    for (uint32_t lr = 0; lr < rows; ++lr) {
      const uint32_t y = ((lr / sr) * (uint32_t)opts->stripe_world + (uint32_t)opts->stripe_rank) * sr + lr % sr;
      std::memcpy(static_cast<unsigned char*>(host) + (size_t)y * w * elem, tmp.data() + (size_t)lr * w * elem, (size_t)w * elem);
    }
    return RT_OK;
  };
  if ((rc = fetch(rgba8, ctx->o_rgba.p, 4)) != RT_OK) return rc;
  if ((rc = fetch(hit_tri, ctx->o_tri.p, 4)) != RT_OK) return rc;
  if ((rc = fetch(hit_t, ctx->o_t.p, 4)) != RT_OK) return rc;
  return RT_OK;
}

extern "C" int rt_deinterleave_device_on(rt_ctx* ctx, void* stream, const void* d_packed, void* d_image, uint32_t width, uint32_t height,
                                         uint32_t elem_bytes, int32_t stripe_rows, int32_t stripe_world, uint32_t rows_per_rank) {
  if (!ctx) return RT_ERR_INVALID;
  if (!d_packed || !d_image || width == 0 || height == 0 || stripe_rows <= 0 || stripe_world < 1 || (elem_bytes != 1 && elem_bytes != 4))
    return ctx->fail(RT_ERR_INVALID, "rt_deinterleave_device: bad argument");
  const uint32_t need = rt_packed_rows(height, stripe_rows, 0, stripe_world);   // rank 0 always holds the most rows
  if (rows_per_rank < need) return ctx->fail(RT_ERR_INVALID, "rt_deinterleave_device: rows_per_rank too small");
  DeviceGuard g(ctx->device);
  rt_launch_deinterleave(d_packed, d_image, width, height, elem_bytes, stripe_rows, stripe_world, rows_per_rank,
                         stream ? static_cast<hipStream_t>(stream) : ctx->stream);
  HIPCHK(hipGetLastError());
  return RT_OK;
}

// Screen rectangle outside of which every pixel of the frame is a miss: what k_cull uses (rt_cull_rect), for the host.
extern "C" int rt_frame_rect(rt_ctx* ctx, const float uniforms[44], int32_t rect[4], uint32_t* miss_rgba) {
  NEED_BVH();
  if (!uniforms || !rect) return ctx->fail(RT_ERR_INVALID, "rt_frame_rect: NULL argument");
  const float fw = uniforms[36], fh = uniforms[37];
  if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw > 65536.0f || fh > 65536.0f) return ctx->fail(RT_ERR_INVALID, "resolution out of range");
  RtRenderParams p;
  std::memset(&p, 0, sizeof(p));
  std::memcpy(p.U, uniforms, sizeof(float) * 44);
  p.width = (uint32_t)fw; p.height = (uint32_t)fh;
  rt_cull_rect(p, ctx->bvh.root_box);
  rect[0] = p.cull_x0; rect[1] = p.cull_y0; rect[2] = p.cull_x1; rect[3] = p.cull_y1;
  if (miss_rgba) *miss_rgba = 0xFF000000u;     // rgba8unorm of (0, 0, 0, 1), compute.wgsl:366 (the value render_frame gives k_cull)
  return RT_OK;
}

extern "C" int rt_assemble_device_on(rt_ctx* ctx, void* stream, const void* d_windows, void* d_image, uint32_t width, uint32_t height,
                                     int32_t stripe_rows, int32_t stripe_world, uint32_t col0, uint32_t n_cols, uint32_t group0,
                                     uint32_t n_rows, uint64_t rank_stride, uint32_t fill) {
  if (!ctx) return RT_ERR_INVALID;
  if (!d_image || width == 0 || height == 0 || stripe_rows <= 0 || stripe_world < 1) return ctx->fail(RT_ERR_INVALID, "rt_assemble_device: bad argument");
  if ((n_cols != 0 && n_rows != 0 && !d_windows) || (uint64_t)col0 + n_cols > width ||
      rank_stride < (uint64_t)n_rows * n_cols)
    return ctx->fail(RT_ERR_INVALID, "rt_assemble_device: the window must lie inside the image and a rank's window must fit its stride");
  DeviceGuard g(ctx->device);
  rt_launch_assemble(d_windows, d_image, width, height, stripe_rows, stripe_world, col0, n_cols, group0, n_rows, (size_t)rank_stride, fill,
                     stream ? static_cast<hipStream_t>(stream) : ctx->stream);
  HIPCHK(hipGetLastError());
  return RT_OK;
}

extern "C" int rt_deinterleave_device(rt_ctx* ctx, const void* d_packed, void* d_image, uint32_t width, uint32_t height,
                                      uint32_t elem_bytes, int32_t stripe_rows, int32_t stripe_world, uint32_t rows_per_rank) {
  return rt_deinterleave_device_on(ctx, nullptr, d_packed, d_image, width, height, elem_bytes, stripe_rows, stripe_world, rows_per_rank);
}
