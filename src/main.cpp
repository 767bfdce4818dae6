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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

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

int main(int argc, char** argv) {
  std::string obj, mtl_dir, out_path, save_bvh, load_bvh;
  uint32_t width = 1920, height = 1080;
  int device = 0, frames = 1;
  bool quad = false, textured = false, v2 = false;
  std::vector<std::pair<int, std::string>> tex_files;
  rt_render_opts opts;
  rt_render_opts_default(&opts);
  rt_camera cam;
  rt_camera_default(&cam, width, height);
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "rt_host: %s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "--obj") obj = next("--obj");
    else if (a == "--mtl-dir") mtl_dir = next("--mtl-dir");
    else if (a == "--width") width = (uint32_t)std::atoi(next("--width"));
    else if (a == "--height") height = (uint32_t)std::atoi(next("--height"));
    else if (a == "--device") device = std::atoi(next("--device"));
    else if (a == "--frames") frames = std::atoi(next("--frames"));
    else if (a == "--spp") opts.gi_samples = std::atoi(next("--spp"));
    else if (a == "--bounces") opts.gi_bounces = std::atoi(next("--bounces"));
    else if (a == "--soft") opts.soft_samples = std::atoi(next("--soft"));
    else if (a == "--stack") opts.max_stack = std::atoi(next("--stack"));
    else if (a == "--out") out_path = next("--out");
    else if (a == "--save-bvh") save_bvh = next("--save-bvh");
    else if (a == "--load-bvh") load_bvh = next("--load-bvh");
    else if (a == "--yaw") cam.yaw = std::atof(next("--yaw"));
    else if (a == "--pitch") cam.pitch = std::atof(next("--pitch"));
    else if (a == "--quad") quad = true;
    else if (a == "--v2") v2 = true;
    else if (a == "--textured") textured = true;
    else if (a == "--texture") {
      std::string t = next("--texture");
      const size_t eq = t.find('=');
      if (eq == std::string::npos) { std::fprintf(stderr, "rt_host: --texture wants slot=file.ppm\n"); return 2; }
      tex_files.emplace_back(std::atoi(t.substr(0, eq).c_str()), t.substr(eq + 1));
    }
    else if (a == "--cam") { for (int k = 0; k < 3; ++k) cam.pos[k] = std::atof(next("--cam")); }
    else if (a == "--mode") {
      std::string m = next("--mode");
      const int layout = opts.layout;
      const int mode = (m == "primary") ? RT_MODE_PRIMARY : (m == "shadow") ? RT_MODE_SHADOW : (m == "frag-web") ? RT_MODE_FRAG_WEB
                     : (m == "frag-root") ? RT_MODE_FRAG_ROOT : (m == "pt") ? RT_MODE_PT : -1;
      if (rt_render_opts_for_mode(&opts, mode) != RT_OK) { std::fprintf(stderr, "rt_host: --mode primary|shadow|pt|frag-web|frag-root\n"); return 2; }
      opts.layout = layout;
    } else if (a == "--layout") {
      std::string m = next("--layout");
      opts.layout = (m == "aos32") ? RT_LAYOUT_AOS32 : RT_LAYOUT_WIDE64;
    } else { std::fprintf(stderr, "rt_host: unknown option %s\n", a.c_str()); return 2; }
  }
  cam.width = width; cam.height = height;

  rt_ctx* ctx = nullptr;
  int rc = rt_create(&ctx, "hip", &device, 1);
  if (rc != RT_OK) die("rt_create", nullptr, rc);
  if (v2 && (rc = rt_set_node_encoding(ctx, RT_NODES_V2)) != RT_OK) die("rt_set_node_encoding", ctx, rc);

  rt_build_stats bs;
  std::memset(&bs, 0, sizeof(bs));
  if (!load_bvh.empty()) {
    if ((rc = rt_load_bvh(ctx, load_bvh.c_str())) != RT_OK) die("rt_load_bvh", ctx, rc);
  } else {
    std::vector<float> pos, nor, tex;
    std::vector<uint32_t> vi, ni, ti;
    std::vector<int32_t> mat_ids;
    std::vector<rt_material_rec> mats;
    if (quad || obj.empty()) {   // generateTestScene(), web/script.js:639-675
      pos = {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0};
      nor = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
      vi = {0, 1, 2, 0, 2, 3}; ni = vi;
    } else {
      rt_mesh* mesh = nullptr;
      char err[512] = {0};
      rc = rt_mesh_load_obj(obj.c_str(), mtl_dir.empty() ? nullptr : mtl_dir.c_str(), &mesh, err, sizeof(err));
      if (rc != RT_OK) { std::fprintf(stderr, "rt_host: %s\n", err); return 1; }
      rt_mesh_view v;
      rt_mesh_view_get(mesh, &v);
      pos.assign(v.positions, v.positions + 3 * (size_t)v.n_vert);
      nor.assign(v.normals, v.normals + 3 * (size_t)v.n_nor);
      vi.assign(v.vert_idx, v.vert_idx + 3 * (size_t)v.n_tri);
      ni.assign(v.nor_idx, v.nor_idx + 3 * (size_t)v.n_tri);
      if (textured) {            // the vt / usemtl data the reference's loadObj drops (script.js:176-196)
        tex.assign(v.texcoords, v.texcoords + 2 * (size_t)v.n_tex);
        ti.assign(v.tex_idx, v.tex_idx + 3 * (size_t)v.n_tri);
        mat_ids.assign(v.material_ids, v.material_ids + v.n_tri);
        const bool explicit_tex = !tex_files.empty();
        for (uint32_t m = 0; m < v.n_materials; ++m) {
          rt_material mm;
          rt_mesh_material(mesh, m, &mm);
          rt_material_rec r;
          r.kd[0] = mm.diffuse[0]; r.kd[1] = mm.diffuse[1]; r.kd[2] = mm.diffuse[2];
          r.texture = -1;
          if (mm.diffuse_texname[0]) {
            if (!explicit_tex) {           // map_Kd file next to the MTL, must be a binary PPM
              const std::string dir = !mtl_dir.empty() ? mtl_dir : (obj.find('/') != std::string::npos ? obj.substr(0, obj.rfind('/')) : std::string("."));
              tex_files.emplace_back((int)m, dir + "/" + mm.diffuse_texname);
            }
            for (auto& tf : tex_files) if (tf.first == (int)m) r.texture = (int)m;
          }
          mats.push_back(r);
        }
      }
      rt_mesh_free(mesh);
      rt_normalize_mesh(pos.data(), (uint32_t)(pos.size() / 3));   // script.js:720
    }
    if ((rc = rt_set_mesh(ctx, pos.data(), (uint32_t)(pos.size() / 3), nor.data(), (uint32_t)(nor.size() / 3), vi.data(), ni.data(),
                          (uint32_t)(vi.size() / 3))) != RT_OK) die("rt_set_mesh", ctx, rc);
    if (textured && !tex.empty() &&
        (rc = rt_set_surface(ctx, tex.data(), (uint32_t)(tex.size() / 2), ti.data(), mat_ids.empty() ? nullptr : mat_ids.data())) != RT_OK)
      die("rt_set_surface", ctx, rc);
    if ((rc = rt_build_lbvh(ctx, &bs)) != RT_OK) die("rt_build_lbvh", ctx, rc);
    if (textured && !tex.empty()) {
      if ((rc = rt_set_materials(ctx, mats.data(), (uint32_t)mats.size())) != RT_OK) die("rt_set_materials", ctx, rc);
      for (auto& tf : tex_files) {
        std::vector<uint8_t> rgb; uint32_t tw = 0, th = 0;
        if (!read_ppm(tf.second, rgb, tw, th)) { std::fprintf(stderr, "rt_host: %s is not a binary PPM (P6, maxval 255)\n", tf.second.c_str()); return 1; }
        if ((rc = rt_set_texture(ctx, (uint32_t)tf.first, rgb.data(), tw, th)) != RT_OK) die("rt_set_texture", ctx, rc);
      }
      opts.flags |= RT_FLAG_TEXTURED;
    }
  }
  if (!save_bvh.empty() && (rc = rt_save_bvh(ctx, save_bvh.c_str())) != RT_OK) die("rt_save_bvh", ctx, rc);

  float U[44];
  if ((rc = rt_make_uniforms(&cam, U)) != RT_OK) die("rt_make_uniforms", ctx, rc);
  std::vector<uint8_t> rgba((size_t)width * height * 4);
  rt_render_stats rs;
  std::memset(&rs, 0, sizeof(rs));
  double ms_sum = 0;
  for (int f = 0; f < frames; ++f) {
    if ((rc = rt_render(ctx, U, &opts, rgba.data(), nullptr, nullptr, &rs)) != RT_OK) die("rt_render", ctx, rc);
    ms_sum += rs.ms_kernel;
  }
  uint32_t n_tri = 0, n_nodes = 0;
  rt_get_counts(ctx, &n_tri, &n_nodes);
  const double rays = (double)(rs.rays_primary + rs.rays_shadow + rs.rays_bounce);
  const double ms = ms_sum / (frames > 0 ? frames : 1);
  std::printf("{\"triangles\": %u, \"nodes\": %u, \"build_ms\": %.4f, \"width\": %u, \"height\": %u, \"mode\": %d, "
              "\"rays\": %.0f, \"kernel_ms\": %.4f, \"mrays_per_s\": %.2f}\n",
              n_tri, n_nodes, bs.ms_total, width, height, opts.mode, rays, ms, ms > 0 ? rays / ms * 1e-3 : 0.0);
  if (!out_path.empty()) {
    FILE* fp = std::fopen(out_path.c_str(), "wb");
    if (!fp) { std::fprintf(stderr, "rt_host: cannot write %s\n", out_path.c_str()); return 1; }
    std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
    // the blit maps texture row 0 to the top of the canvas (v = 0 at clip y = +1, script.js:628)
    for (uint32_t y = 0; y < height; ++y) {
      const uint8_t* row = rgba.data() + (size_t)y * width * 4;
      for (uint32_t x = 0; x < width; ++x) std::fwrite(row + 4 * x, 1, 3, fp);
    }
    std::fclose(fp);
  }
  rt_destroy(ctx);
  return 0;
}
