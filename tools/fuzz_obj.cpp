// tools/fuzz_obj.cpp — mutation smoke of the OBJ / MTL parser under host sanitizers (no GPU code involved):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/fuzz_obj.cpp raytracer_amd/csrc/rt_scene.cpp -o /tmp/fuzz_obj
//   gunzip -c tests/golden/plane.obj.gz > /tmp/plane.obj && /tmp/fuzz_obj /tmp/plane.obj      (300 mutated files: "ok 41 rejected 259", no report)
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
      switch (rng() % 6) {
        case 0: s[pos] = (char)(rng() % 256); break;
        case 1: s.insert(pos, "f 1/2/3 999999999/1/1 -5//-7 4\n"); break;
        case 2: s.insert(pos, "v 1e999 nan -inf\nvt\nvn 1\nf\nf 1\nf 1 2\nusemtl \nmtllib nope.mtl\n"); break;
        case 3: s.erase(pos, rng() % 50); break;
        case 4: s.insert(pos, std::string(rng() % 300, '7')); break;
        case 5: s.insert(pos, "f 0/0/0 0 0\nf -999999 1 2\nf 1//2 3//4 5//6 7//8 9//10\n"); break;
      }
    }
    std::ofstream o("/tmp/fuzz_obj_case.obj", std::ios::binary); o << s; o.close();
    rt_mesh* m = nullptr; char err[256] = {0};
    int rc = rt_mesh_load_obj("/tmp/fuzz_obj_case.obj", nullptr, &m, err, sizeof err);
    if (rc == 0 && m) { rt_mesh_view v; rt_mesh_view_get(m, &v); ok++; rt_mesh_free(m); } else bad++;
  }
  printf("ok %d rejected %d\n", ok, bad);
}
