// tools/fuzz_obj.cpp — mutation smoke of the OBJ / MTL parser under host sanitizers (no GPU code involved):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/fuzz_obj.cpp raytracer_amd/csrc/rt_scene.cpp -o /tmp/fuzz_obj
//   gunzip -c tests/golden/plane.obj.gz > /tmp/plane.obj && /tmp/fuzz_obj /tmp/plane.obj      (300 mutated OBJ files incl. n-gons for the ear clipping + 300 MTL texts: "ok 346 rejected 254", no report)
#include "rtcore.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <fstream>
#include <random>
int main(int argc, char** argv) {
  std::ifstream f(argv[1], std::ios::binary); std::string base((std::istreambuf_iterator<char>(f)), {});
  std::mt19937 rng(7);
  int ok = 0, bad = 0;
  for (int it = 0; it < 300; ++it) {
    std::string s = base.substr(0, 20000 + rng() % 20000);
    int nmut = 1 + rng() % 40;
    for (int k = 0; k < nmut; ++k) {
      size_t pos = rng() % s.size();
      switch (rng() % 9) {
        case 0: s[pos] = (char)(rng() % 256); break;
        case 1: s.insert(pos, "f 1/2/3 999999999/1/1 -5//-7 4\n"); break;
        case 2: s.insert(pos, "v 1e999 nan -inf\nvt\nvn 1\nf\nf 1\nf 1 2\nusemtl \nmtllib nope.mtl\n"); break;
        case 3: s.erase(pos, rng() % 50); break;
        case 4: s.insert(pos, std::string(rng() % 300, '7')); break;
        case 5: s.insert(pos, "f 0/0/0 0 0\nf -999999 1 2\nf 1//2 3//4 5//6 7//8 9//10\n"); break;
        case 6: s.insert(pos, "f 1 2 3 4 5 6 7\nf 1 1 1 1 1 1\nf 3 2 1 3 2 1 3 2 1\nf -1 -2 -3 -4 -5 -6 -7 -8 -9 -10 -11 -12\n"); break;   // ear clipping: n-gons, degenerate rings
        case 7: { std::string poly = "f"; int n = 5 + rng() % 60; for (int q = 0; q < n; ++q) poly += " " + std::to_string(1 + rng() % 4000); s.insert(pos, poly + "\n"); } break;
        case 8: s.insert(pos, "v nan nan nan\nv 1e38 -1e38 1e38\nv 0 0 0\nv 0 0 0\nv 0 0 0\nf -1 -2 -3 -4 -5\n"); break;
      }
    }
    std::ofstream o("/tmp/fuzz_obj_case.obj", std::ios::binary); o << s; o.close();
    rt_mesh* m = nullptr; char err[256] = {0};
    int rc = rt_mesh_load_obj("/tmp/fuzz_obj_case.obj", nullptr, &m, err, sizeof err);
    if (rc == 0 && m) { rt_mesh_view v; rt_mesh_view_get(m, &v); ok++; rt_mesh_free(m); } else bad++;
  }
  // MTL lines: unknown parameters, tabs, missing values, keys before any newmtl, very long values
  const std::string obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n";
  for (int it = 0; it < 300; ++it) {
    std::string mtl = "stray 1\nnewmtl a\nKd 1 0 0\n";
    int n = rng() % 30;
    for (int k = 0; k < n; ++k) {
      switch (rng() % 7) {
        case 0: mtl += "Pr\t0.5\n"; break;
        case 1: mtl += "key" + std::to_string(rng() % 5) + " " + std::string(rng() % 600, 'x') + "\n"; break;
        case 2: mtl += "\t\t  odd\tkey with  spaces \t \n"; break;
        case 3: mtl += "novalue\n"; break;
        case 4: mtl += std::string(1, (char)(rng() % 256)) + " z\n"; break;
        case 5: mtl += "newmtl m" + std::to_string(rng() % 3) + "\nmap_Kd " + std::string(rng() % 400, 't') + ".png\n"; break;
        case 6: mtl += "Kd nan\nd\nTr x\nillum\n"; break;
      }
    }
    rt_mesh* m = nullptr; char err[256] = {0};
    int rc = rt_mesh_parse_obj(obj.data(), obj.size(), mtl.data(), mtl.size(), &m, err, sizeof err);
    if (rc == 0 && m) {
      rt_mesh_view v; rt_mesh_view_get(m, &v);
      for (uint32_t i = 0; i < v.n_materials; ++i)
        for (uint32_t k = 0; k < rt_mesh_material_n_unknown(m, i); ++k) { const char *key, *val; rt_mesh_material_unknown(m, i, k, &key, &val); (void)key[0]; (void)val[0]; }
      ok++; rt_mesh_free(m);
    } else bad++;
  }
  printf("ok %d rejected %d\n", ok, bad);
}
