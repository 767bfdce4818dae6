// src/main.cpp — the repo's own C++ host (role of the reference's src/main.cpp + Makefile,
// which is a merge-conflicted OpenGL stub that never traces a ray: SURVEY.md F6).
//
// Drives librtcore.so through the C ABI of include/rtcore.h only:
//   OBJ/MTL -> normalizeMesh -> GPU LBVH build -> one frame() -> PPM image + one JSON line.
// Mirrors the page-lifetime flow of web/script.js:717-748 (load, normalise, build, upload)
// and frame() :865-945 (uniforms, dispatch).
//
//   rt_host --obj web/plane.obj [--mtl-dir models] [--width 1920 --height 1080]
//           [--mode primary|shadow|pt|frag-web|frag-root] [--spp 32] [--bounces 3] [--soft 4] [--stack 12]
//             (--mode sets the sample counts / stack depth its shader file is written with: give --spp etc. AFTER it;
//              frag-web / frag-root = the older fragment shaders web/shader.wgsl / shader.wgsl, sky on a miss)
//           [--layout wide64|aos32] [--cam x y z] [--yaw r] [--pitch r] [--frames n]
//           [--save-bvh f] [--load-bvh f] [--out image.ppm] [--device 0] [--quad] [--v2]
//           [--textured] [--texture slot=image.ppm ...]   materials from the MTL (Kd, map_Kd -> slot = material index
//           unless --texture assigns slots explicitly); textures are binary PPM (P6) files, the image has no JPEG/PNG decoder
//           [--gpus N [--batch B]]   one PROCESS per GPU, forked before the first HIP call; RCCL on rt_stream():
//             rank 0 loads + builds, the flat BVH travels as one blob (grouped ncclSend/ncclRecv scatter + in-place
//             ncclAllGather), every rank renders its 16-row stripes of B frames per pipeline (rt_render_batch_device) into
//             packed buffers, a grouped ncclSend/ncclRecv gathers them on rank 0, rt_deinterleave_device assembles the frames.
//             N = 1 runs the same code with a communicator of one (tests/test_gpu_host.py).
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "rtcore.h"

// binary PPM (P6, maxval 255) -> RGB8 rows, top row first
static bool read_ppm(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0};
  unsigned vals[3] = {0, 0, 0};
  int got = 0;
  bool ok = std::fscanf(f, "%2s", magic) == 1 && std::strcmp(magic, "P6") == 0;
  while (ok && got < 3) {
    int c = std::fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
    if (c == EOF) { ok = false; break; }
    if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
    std::ungetc(c, f);
    if (std::fscanf(f, "%u", &vals[got]) != 1) { ok = false; break; }
    got += 1;
  }
  if (ok) std::fgetc(f);   // the single whitespace byte after maxval
  ok = ok && vals[0] > 0 && vals[1] > 0 && vals[0] <= 32768 && vals[1] <= 32768 && vals[2] == 255;
  if (ok) {
    w = vals[0]; h = vals[1];
    rgb.resize((size_t)w * h * 3);
    ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
  }
  std::fclose(f);
  return ok;
}

static void die(const char* what, rt_ctx* ctx, int rc) {
  std::fprintf(stderr, "rt_host: %s failed (%d): %s\n", what, rc, rt_last_error(ctx));
  std::exit(1);
}

struct Options {
  std::string obj, mtl_dir, out_path, save_bvh, load_bvh;
  uint32_t width = 1920, height = 1080;
  int device = 0, frames = 1, gpus = 0, batch = RT_MAX_BATCH;
  bool quad = false, textured = false, v2 = false;
  std::vector<std::pair<int, std::string>> tex_files;
  rt_render_opts opts;
  rt_camera cam;
};

// material table of a loaded OBJ: Kd + texture slot (= material index) for every map_Kd that has a texture file
static void materials_of(Options& o, rt_mesh* mesh, const rt_mesh_view& v, std::vector<rt_material_rec>& mats) {
  const bool explicit_tex = !o.tex_files.empty();
  for (uint32_t m = 0; m < v.n_materials; ++m) {
    rt_material mm;
    rt_mesh_material(mesh, m, &mm);
    rt_material_rec r;
    r.kd[0] = mm.diffuse[0]; r.kd[1] = mm.diffuse[1]; r.kd[2] = mm.diffuse[2];
    r.texture = -1;
    if (mm.diffuse_texname[0]) {
      if (!explicit_tex) {           // map_Kd file next to the MTL, must be a binary PPM
        const std::string dir = !o.mtl_dir.empty() ? o.mtl_dir : (o.obj.find('/') != std::string::npos ? o.obj.substr(0, o.obj.rfind('/')) : std::string("."));
        o.tex_files.emplace_back((int)m, dir + "/" + mm.diffuse_texname);
      }
      for (auto& tf : o.tex_files) if (tf.first == (int)m) r.texture = (int)m;
    }
    mats.push_back(r);
  }
}

// OBJ/MTL (or the built-in quad) -> normalised mesh -> rt_set_mesh (+ surface, materials, textures) -> rt_build_lbvh,
// or a cached BVH.  Sets RT_FLAG_TEXTURED in o.opts when the scene has surface data.  (script.js:717-748)
static void load_and_build(Options& o, rt_ctx* ctx, rt_build_stats& bs, std::vector<rt_material_rec>& mats) {
  int rc;
  if (!o.load_bvh.empty()) {
    if ((rc = rt_load_bvh(ctx, o.load_bvh.c_str())) != RT_OK) die("rt_load_bvh", ctx, rc);
    return;
  }
  std::vector<float> pos, nor, tex;
  std::vector<uint32_t> vi, ni, ti;
  std::vector<int32_t> mat_ids;
  if (o.quad || o.obj.empty()) {   // generateTestScene(), web/script.js:639-675
    pos = {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0};
    nor = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
    vi = {0, 1, 2, 0, 2, 3}; ni = vi;
  } else {
    rt_mesh* mesh = nullptr;
    char err[512] = {0};
    rc = rt_mesh_load_obj(o.obj.c_str(), o.mtl_dir.empty() ? nullptr : o.mtl_dir.c_str(), &mesh, err, sizeof(err));
    if (rc != RT_OK) { std::fprintf(stderr, "rt_host: %s\n", err); std::exit(1); }
    rt_mesh_view v;
    rt_mesh_view_get(mesh, &v);
    pos.assign(v.positions, v.positions + 3 * (size_t)v.n_vert);
    nor.assign(v.normals, v.normals + 3 * (size_t)v.n_nor);
    vi.assign(v.vert_idx, v.vert_idx + 3 * (size_t)v.n_tri);
    ni.assign(v.nor_idx, v.nor_idx + 3 * (size_t)v.n_tri);
    if (o.textured) {            // the vt / usemtl data the reference's loadObj drops (script.js:176-196)
      tex.assign(v.texcoords, v.texcoords + 2 * (size_t)v.n_tex);
      ti.assign(v.tex_idx, v.tex_idx + 3 * (size_t)v.n_tri);
      mat_ids.assign(v.material_ids, v.material_ids + v.n_tri);
      materials_of(o, mesh, v, mats);
    }
    rt_mesh_free(mesh);
    rt_normalize_mesh(pos.data(), (uint32_t)(pos.size() / 3));   // script.js:720
  }
  if ((rc = rt_set_mesh(ctx, pos.data(), (uint32_t)(pos.size() / 3), nor.data(), (uint32_t)(nor.size() / 3), vi.data(), ni.data(),
                        (uint32_t)(vi.size() / 3))) != RT_OK) die("rt_set_mesh", ctx, rc);
  if (o.textured && !tex.empty() &&
      (rc = rt_set_surface(ctx, tex.data(), (uint32_t)(tex.size() / 2), ti.data(), mat_ids.empty() ? nullptr : mat_ids.data())) != RT_OK)
    die("rt_set_surface", ctx, rc);
  if ((rc = rt_build_lbvh(ctx, &bs)) != RT_OK) die("rt_build_lbvh", ctx, rc);
  if (o.textured && !tex.empty()) o.opts.flags |= RT_FLAG_TEXTURED;
}

// materials + textures: small host-side descriptions, set by every rank itself
static void bind_surface(const Options& o, rt_ctx* ctx, const std::vector<rt_material_rec>& mats) {
  int rc;
  if (!(o.opts.flags & RT_FLAG_TEXTURED)) return;
  if ((rc = rt_set_materials(ctx, mats.data(), (uint32_t)mats.size())) != RT_OK) die("rt_set_materials", ctx, rc);
  for (auto& tf : o.tex_files) {
    std::vector<uint8_t> rgb; uint32_t tw = 0, th = 0;
    if (!read_ppm(tf.second, rgb, tw, th)) { std::fprintf(stderr, "rt_host: %s is not a binary PPM (P6, maxval 255)\n", tf.second.c_str()); std::exit(1); }
    if ((rc = rt_set_texture(ctx, (uint32_t)tf.first, rgb.data(), tw, th)) != RT_OK) die("rt_set_texture", ctx, rc);
  }
}

static void write_ppm(const std::string& path, const uint8_t* rgba, uint32_t width, uint32_t height) {
  FILE* fp = std::fopen(path.c_str(), "wb");
  if (!fp) { std::fprintf(stderr, "rt_host: cannot write %s\n", path.c_str()); std::exit(1); }
  std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
  // the blit maps texture row 0 to the top of the canvas (v = 0 at clip y = +1, script.js:628)
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t* row = rgba + (size_t)y * width * 4;
    for (uint32_t x = 0; x < width; ++x) std::fwrite(row + 4 * x, 1, 3, fp);
  }
  std::fclose(fp);
}

// ---------------------------------------------------------------------------------------------- one process, one GPU
static int run_single(Options& o) {
  rt_ctx* ctx = nullptr;
  int rc = rt_create(&ctx, "hip", &o.device, 1);
  if (rc != RT_OK) die("rt_create", nullptr, rc);
  if (o.v2 && (rc = rt_set_node_encoding(ctx, RT_NODES_V2)) != RT_OK) die("rt_set_node_encoding", ctx, rc);
  rt_build_stats bs;
  std::memset(&bs, 0, sizeof(bs));
  std::vector<rt_material_rec> mats;
  load_and_build(o, ctx, bs, mats);
  bind_surface(o, ctx, mats);
  if (!o.save_bvh.empty() && (rc = rt_save_bvh(ctx, o.save_bvh.c_str())) != RT_OK) die("rt_save_bvh", ctx, rc);

  float U[44];
  if ((rc = rt_make_uniforms(&o.cam, U)) != RT_OK) die("rt_make_uniforms", ctx, rc);
  std::vector<uint8_t> rgba((size_t)o.width * o.height * 4);
  rt_render_stats rs;
  std::memset(&rs, 0, sizeof(rs));
  double ms_sum = 0;
  for (int f = 0; f < o.frames; ++f) {
    if ((rc = rt_render(ctx, U, &o.opts, rgba.data(), nullptr, nullptr, &rs)) != RT_OK) die("rt_render", ctx, rc);
    ms_sum += rs.ms_kernel;
  }
  uint32_t n_tri = 0, n_nodes = 0;
  rt_get_counts(ctx, &n_tri, &n_nodes);
  const double rays = (double)(rs.rays_primary + rs.rays_shadow + rs.rays_bounce);
  const double ms = ms_sum / (o.frames > 0 ? o.frames : 1);
  std::printf("{\"triangles\": %u, \"nodes\": %u, \"build_ms\": %.4f, \"width\": %u, \"height\": %u, \"mode\": %d, "
              "\"rays\": %.0f, \"kernel_ms\": %.4f, \"mrays_per_s\": %.2f}\n",
              n_tri, n_nodes, bs.ms_total, o.width, o.height, o.opts.mode, rays, ms, ms > 0 ? rays / ms * 1e-3 : 0.0);
  if (!o.out_path.empty()) write_ppm(o.out_path, rgba.data(), o.width, o.height);
  rt_destroy(ctx);
  return 0;
}

// ---------------------------------------------------------------------------------------------- one process per GPU
#define HIPOK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "rt_host[%d]: %s: %s\n", rank, #expr, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define NCCLOK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { std::fprintf(stderr, "rt_host[%d]: %s: %s\n", rank, #expr, ncclGetErrorString(r_)); std::exit(1); } } while (0)
#define RTOK(expr) do { int rc_ = (expr); if (rc_ != RT_OK) { std::fprintf(stderr, "rt_host[%d]: ", rank); die(#expr, ctx, rc_); } } while (0)

struct Shared {                 // mapped before the fork: how rank 0's communicator id reaches the other ranks
  ncclUniqueId id;
  volatile int id_ready;
};

static constexpr int STRIPE_ROWS = 16;   // stripe s -> rank s % world (raytracer_amd/dist.py)

static int run_rank(Options o, int rank, int world, Shared* sh) {
  // nothing has touched HIP in this process (nor in the parent) before this point
  const int device = o.device + rank;
  HIPOK(hipSetDevice(device));
  if (rank == 0) {
    NCCLOK(ncclGetUniqueId(&sh->id));
    __sync_synchronize();
    sh->id_ready = 1;
  } else {
    while (!sh->id_ready) usleep(1000);
    __sync_synchronize();
  }
  ncclUniqueId id;
  std::memcpy(&id, &sh->id, sizeof(id));
  ncclComm_t comm;
  NCCLOK(ncclCommInitRank(&comm, world, id, rank));

  rt_ctx* ctx = nullptr;
  { const int rc = rt_create(&ctx, "hip", &device, 1); if (rc != RT_OK) die("rt_create", nullptr, rc); }
  hipStream_t s = static_cast<hipStream_t>(rt_stream(ctx));        // every RCCL call below is ordered on the context stream
  if (o.v2) RTOK(rt_set_node_encoding(ctx, RT_NODES_V2));

  // ---- hot path 1 on rank 0, then replication
  rt_build_stats bs;
  std::memset(&bs, 0, sizeof(bs));
  std::vector<rt_material_rec> mats;
  unsigned long long meta_h[8] = {0};                              // T, N, encoding, has_uv, flags, n_materials
  if (rank == 0) {
    load_and_build(o, ctx, bs, mats);
    if (!o.save_bvh.empty()) RTOK(rt_save_bvh(ctx, o.save_bvh.c_str()));
    uint32_t T = 0, N = 0;
    RTOK(rt_get_counts(ctx, &T, &N));
    void* uvp = nullptr; uint64_t uvb = 0;
    RTOK(rt_tri_uv_device(ctx, &uvp, &uvb));
    meta_h[0] = T; meta_h[1] = N; meta_h[2] = (unsigned long long)rt_get_node_encoding(ctx); meta_h[3] = uvb ? 1 : 0;
    meta_h[4] = (unsigned long long)o.opts.flags; meta_h[5] = mats.size();
  }
  unsigned long long* meta_d = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&meta_d), sizeof(meta_h)));
  HIPOK(hipMemcpyAsync(meta_d, meta_h, sizeof(meta_h), hipMemcpyHostToDevice, s));
  NCCLOK(ncclBroadcast(meta_d, meta_d, 8, ncclUint64, 0, comm, s));
  HIPOK(hipMemcpyAsync(meta_h, meta_d, sizeof(meta_h), hipMemcpyDeviceToHost, s));
  HIPOK(hipStreamSynchronize(s));
  const uint32_t T = (uint32_t)meta_h[0], N = (uint32_t)meta_h[1];
  const bool has_uv = meta_h[3] != 0;
  o.opts.flags = (uint32_t)meta_h[4];
  // one blob {nodes | triPos | triNor | surface records}, padded to world equal chunks
  const uint64_t seg[4] = {32ull * N, 48ull * T, 48ull * T, has_uv ? 32ull * T : 0ull};
  uint64_t off[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 4; ++k) off[k + 1] = off[k] + seg[k];
  const uint64_t align = 256ull * (uint64_t)world, total = (off[4] + align - 1) / align * align, chunk = total / (uint64_t)world;
  unsigned char* blob = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&blob), total));
  rt_bvh_view view;
  view.d_nodes = blob + off[0]; view.nodes_bytes = seg[0];
  view.d_tri_pos = blob + off[1]; view.tri_pos_bytes = seg[1];
  view.d_tri_nor = blob + off[2]; view.tri_nor_bytes = seg[2];
  view.n_nodes = N; view.n_tri = T;
  const auto t_rep0 = std::chrono::steady_clock::now();
  if (rank == 0) {
    RTOK(rt_bvh_export_device(ctx, &view));
    if (has_uv) RTOK(rt_tri_uv_export_device(ctx, blob + off[3], seg[3]));
  }
  if (world > 1) {
    // scatter: rank 0 sends chunk r to rank r (all of rank 0's links at once) ...
    NCCLOK(ncclGroupStart());
    if (rank == 0) for (int r = 1; r < world; ++r) NCCLOK(ncclSend(blob + (uint64_t)r * chunk, chunk, ncclUint8, r, comm, s));
    else NCCLOK(ncclRecv(blob + (uint64_t)rank * chunk, chunk, ncclUint8, 0, comm, s));
    NCCLOK(ncclGroupEnd());
    // ... all-gather, in place: every rank ends up with the blob
    NCCLOK(ncclAllGather(blob + (uint64_t)rank * chunk, blob, chunk, ncclUint8, comm, s));
  }
  HIPOK(hipStreamSynchronize(s));
  const double replicate_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rep0).count();
  if (rank != 0) {
    RTOK(rt_set_node_encoding(ctx, (int)meta_h[2]));
    RTOK(rt_bvh_adopt_device(ctx, &view));                          // the blob stays alive until the context is gone
    if (has_uv) RTOK(rt_adopt_tri_uv_device(ctx, blob + off[3], T));
    if (o.opts.flags & RT_FLAG_TEXTURED) {                          // materials come from the MTL: every rank parses it itself
      rt_mesh* mesh = nullptr; char err[512] = {0};
      if (!o.obj.empty() && rt_mesh_load_obj(o.obj.c_str(), o.mtl_dir.empty() ? nullptr : o.mtl_dir.c_str(), &mesh, err, sizeof(err)) == RT_OK) {
        rt_mesh_view v; rt_mesh_view_get(mesh, &v);
        materials_of(o, mesh, v, mats);
        rt_mesh_free(mesh);
      }
    }
  }
  bind_surface(o, ctx, mats);

  // ---- hot path 2: my stripes of every frame, B frames per pipeline
  float U[44];
  RTOK(rt_make_uniforms(&o.cam, U));
  rt_render_opts ro = o.opts;
  ro.stripe_rows = world > 1 ? STRIPE_ROWS : 0; ro.stripe_rank = rank; ro.stripe_world = world;
  const uint32_t W = o.width, H = o.height;
  const uint32_t rpr = world > 1 ? rt_packed_rows(H, STRIPE_ROWS, 0, world) : H;      // rank 0 always owns the most rows
  const int B = o.batch < 1 ? 1 : (o.batch > RT_MAX_BATCH ? RT_MAX_BATCH : o.batch);
  const size_t share = (size_t)rpr * W;                                               // pixels of one packed share
  uint32_t *packed = nullptr, *staging = nullptr, *gathered = nullptr, *images = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&packed), sizeof(uint32_t) * share * B));
  HIPOK(hipMalloc(reinterpret_cast<void**>(&staging), sizeof(uint32_t) * share * B));
  HIPOK(hipMemsetAsync(packed, 0, sizeof(uint32_t) * share * B, s));
  if (rank == 0) {
    HIPOK(hipMalloc(reinterpret_cast<void**>(&gathered), sizeof(uint32_t) * share * B * world));
    HIPOK(hipMalloc(reinterpret_cast<void**>(&images), sizeof(uint32_t) * (size_t)W * H * B));
  }
  // rays of one frame (all ranks' shares): one counted launch each, summed over the ranks
  rt_render_opts co = ro;
  co.flags |= RT_FLAG_COUNT_WORK;
  rt_render_stats rs;
  RTOK(rt_render_device(ctx, U, &co, packed, nullptr, nullptr, &rs));
  unsigned long long rays_h = rs.rays_primary + rs.rays_shadow + rs.rays_bounce;
  HIPOK(hipMemcpyAsync(meta_d, &rays_h, sizeof(rays_h), hipMemcpyHostToDevice, s));
  NCCLOK(ncclAllReduce(meta_d, meta_d, 1, ncclUint64, ncclSum, comm, s));
  HIPOK(hipMemcpyAsync(&rays_h, meta_d, sizeof(rays_h), hipMemcpyDeviceToHost, s));
  HIPOK(hipStreamSynchronize(s));

  std::vector<float> Ub((size_t)44 * B);
  for (int f = 0; f < B; ++f) std::memcpy(Ub.data() + 44 * (size_t)f, U, sizeof(U));
  void* outs[RT_MAX_BATCH];
  const auto t0 = std::chrono::steady_clock::now();
  for (int f0 = 0; f0 < o.frames; f0 += B) {
    const int n = o.frames - f0 < B ? o.frames - f0 : B;
    for (int f = 0; f < n; ++f) outs[f] = packed + share * (size_t)f;
    RTOK(rt_render_batch_device(ctx, (uint32_t)n, Ub.data(), &ro, outs, nullptr, nullptr));
    // Only the WINDOW of the batch crosses the links (rtcore.h, rt_frame_rect): the union of the frames' screen rectangles,
    // widened to 16-byte columns and to whole stripe groups (one stripe of every rank) so that it is the same on all ranks.
    int32_t x0 = (int32_t)W, y0 = (int32_t)H, x1 = -1, y1 = -1;
    uint32_t miss = 0;
    for (int f = 0; f < n; ++f) {
      int32_t r[4];
      RTOK(rt_frame_rect(ctx, Ub.data() + 44 * (size_t)f, r, &miss));
      if (r[2] >= r[0] && r[3] >= r[1]) { x0 = std::min(x0, r[0]); y0 = std::min(y0, r[1]); x1 = std::max(x1, r[2]); y1 = std::max(y1, r[3]); }
    }
    uint32_t col0 = 0, n_cols = 0, group0 = 0, n_rows = 0;
    if (x1 >= x0 && y1 >= y0) {
      col0 = (uint32_t)std::max(x0, 0) & ~3u;
      n_cols = std::min(W - col0, ((uint32_t)std::min<int32_t>(x1, (int32_t)W - 1) + 1u - col0 + 3u) & ~3u);
      const uint32_t rows_per_group = (uint32_t)STRIPE_ROWS * (uint32_t)world;
      group0 = (uint32_t)std::max(y0, 0) / rows_per_group;
      const uint32_t group1 = (uint32_t)std::min<int32_t>(y1, (int32_t)H - 1) / rows_per_group + 1u;
      n_rows = std::min(group1 * (uint32_t)STRIPE_ROWS, rpr) - group0 * (uint32_t)STRIPE_ROWS;
    }
    const size_t win = (size_t)n_rows * n_cols;                       // elements of one frame's window of one rank
    if (win) {
      for (int f = 0; f < n; ++f)
        HIPOK(hipMemcpy2DAsync(staging + win * (size_t)f, sizeof(uint32_t) * n_cols,
                               packed + share * (size_t)f + (size_t)group0 * STRIPE_ROWS * W + col0, sizeof(uint32_t) * W,
                               sizeof(uint32_t) * n_cols, n_rows, hipMemcpyDeviceToDevice, s));
      // gather of the batch on rank 0: rank r's n windows land at gathered[r][0 .. n)
      NCCLOK(ncclGroupStart());
      if (rank == 0) for (int r = 0; r < world; ++r) NCCLOK(ncclRecv(gathered + win * (size_t)n * r, win * n, ncclUint32, r, comm, s));
      NCCLOK(ncclSend(staging, win * n, ncclUint32, 0, comm, s));
      NCCLOK(ncclGroupEnd());
    }
    if (rank == 0)
      for (int f = 0; f < n; ++f)
        RTOK(rt_assemble_device_on(ctx, nullptr, win ? gathered + win * (size_t)f : nullptr, images + (size_t)W * H * f, W, H, STRIPE_ROWS, world,
                                   col0, n_cols, group0, n_rows, win * (size_t)n, miss));
  }
  HIPOK(hipStreamSynchronize(s));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (o.frames > 0 ? o.frames : 1);

  if (rank == 0) {
    std::printf("{\"triangles\": %u, \"nodes\": %u, \"build_ms\": %.4f, \"replicate_ms\": %.3f, \"width\": %u, \"height\": %u, \"mode\": %d, "
                "\"n_gpus\": %d, \"frames\": %d, \"frames_per_pipeline\": %d, \"rays\": %llu, \"ms_per_frame\": %.4f, \"mrays_per_s\": %.2f}\n",
                T, N, bs.ms_total, replicate_ms, W, H, o.opts.mode, world, o.frames, B, rays_h, ms, ms > 0 ? (double)rays_h / ms * 1e-3 : 0.0);
    if (!o.out_path.empty() && o.frames > 0) {
      const int last = (o.frames - 1) % B;
      std::vector<uint8_t> rgba((size_t)W * H * 4);
      HIPOK(hipMemcpy(rgba.data(), images + (size_t)W * H * last, rgba.size(), hipMemcpyDeviceToHost));
      write_ppm(o.out_path, rgba.data(), W, H);
    }
  }
  rt_destroy(ctx);
  NCCLOK(ncclCommDestroy(comm));
  (void)hipFree(meta_d); (void)hipFree(blob); (void)hipFree(packed); (void)hipFree(staging); (void)hipFree(gathered); (void)hipFree(images);
  return 0;
}

static int run_multi(const Options& o) {
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);                   // dmabuf IPC (RCCL across processes needs it on this driver)
  Shared* sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (sh == MAP_FAILED) { std::perror("rt_host: mmap"); return 1; }
  std::memset(sh, 0, sizeof(*sh));
  std::fflush(nullptr);
  std::vector<pid_t> kids;
  for (int r = 0; r < o.gpus; ++r) {
    const pid_t pid = fork();                                      // before ANY HIP call: a forked HIP runtime is unusable
    if (pid < 0) { std::perror("rt_host: fork"); return 1; }
    if (pid == 0) { const int rc = run_rank(o, r, o.gpus, sh); std::fflush(nullptr); _exit(rc); }
    kids.push_back(pid);
  }
  int worst = 0;
  for (pid_t k : kids) {
    int st = 0;
    if (waitpid(k, &st, 0) < 0) { worst = 1; continue; }
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (rc > worst) worst = rc;
  }
  munmap(sh, sizeof(Shared));
  return worst;
}

int main(int argc, char** argv) {
  Options o;
  rt_render_opts_default(&o.opts);
  rt_camera_default(&o.cam, o.width, o.height);
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "rt_host: %s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "--obj") o.obj = next("--obj");
    else if (a == "--mtl-dir") o.mtl_dir = next("--mtl-dir");
    else if (a == "--width") o.width = (uint32_t)std::atoi(next("--width"));
    else if (a == "--height") o.height = (uint32_t)std::atoi(next("--height"));
    else if (a == "--device") o.device = std::atoi(next("--device"));
    else if (a == "--gpus") o.gpus = std::atoi(next("--gpus"));
    else if (a == "--batch") o.batch = std::atoi(next("--batch"));
    else if (a == "--frames") o.frames = std::atoi(next("--frames"));
    else if (a == "--spp") o.opts.gi_samples = std::atoi(next("--spp"));
    else if (a == "--bounces") o.opts.gi_bounces = std::atoi(next("--bounces"));
    else if (a == "--soft") o.opts.soft_samples = std::atoi(next("--soft"));
    else if (a == "--stack") o.opts.max_stack = std::atoi(next("--stack"));
    else if (a == "--out") o.out_path = next("--out");
    else if (a == "--save-bvh") o.save_bvh = next("--save-bvh");
    else if (a == "--load-bvh") o.load_bvh = next("--load-bvh");
    else if (a == "--yaw") o.cam.yaw = std::atof(next("--yaw"));
    else if (a == "--pitch") o.cam.pitch = std::atof(next("--pitch"));
    else if (a == "--quad") o.quad = true;
    else if (a == "--v2") o.v2 = true;
    else if (a == "--textured") o.textured = true;
    else if (a == "--texture") {
      std::string t = next("--texture");
      const size_t eq = t.find('=');
      if (eq == std::string::npos) { std::fprintf(stderr, "rt_host: --texture wants slot=file.ppm\n"); return 2; }
      o.tex_files.emplace_back(std::atoi(t.substr(0, eq).c_str()), t.substr(eq + 1));
    }
    else if (a == "--cam") { for (int k = 0; k < 3; ++k) o.cam.pos[k] = std::atof(next("--cam")); }
    else if (a == "--mode") {
      std::string m = next("--mode");
      const int layout = o.opts.layout;
      const int mode = (m == "primary") ? RT_MODE_PRIMARY : (m == "shadow") ? RT_MODE_SHADOW : (m == "frag-web") ? RT_MODE_FRAG_WEB
                     : (m == "frag-root") ? RT_MODE_FRAG_ROOT : (m == "pt") ? RT_MODE_PT : -1;
      if (rt_render_opts_for_mode(&o.opts, mode) != RT_OK) { std::fprintf(stderr, "rt_host: --mode primary|shadow|pt|frag-web|frag-root\n"); return 2; }
      o.opts.layout = layout;
    } else if (a == "--layout") {
      std::string m = next("--layout");
      o.opts.layout = (m == "aos32") ? RT_LAYOUT_AOS32 : RT_LAYOUT_WIDE64;
    } else { std::fprintf(stderr, "rt_host: unknown option %s\n", a.c_str()); return 2; }
  }
  o.cam.width = o.width; o.cam.height = o.height;
  if (o.gpus < 0 || o.gpus > 16) { std::fprintf(stderr, "rt_host: --gpus 1 .. 16\n"); return 2; }
  if (o.gpus >= 1 && o.opts.mode > RT_MODE_PT) { std::fprintf(stderr, "rt_host: --gpus renders through rt_render_batch_device: --mode primary|shadow|pt\n"); return 2; }
  return o.gpus >= 1 ? run_multi(o) : run_single(o);
}
