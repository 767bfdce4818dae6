// rt_scene.cpp — host-side scene input of librtcore: OBJ/MTL ingest, mesh normalisation and
// the 176-byte uniform block.  Pure C++ (no HIP); usable without a GPU.
//
// Reference behaviour restated here:
//   OBJ/MTL   tiny_obj_loader v2.0.0 (/root/reference/tiny_obj_loader.h): float grammar and
//             arithmetic of tryParseDouble :954-1122, index fix-up :867-907, face triples
//             :1312-1372, quad split along the shorter diagonal :1701-1813, the built-in ear
//             clipping of polygons with more than 4 corners :1945-2191 (+ pnpoly :1616-1628),
//             material keys :188-247 and unknown_parameter :2749-2762.
//   normalizeMesh  web/script.js:207-227
//   uniforms       web/script.js:30-154, 591-612, 880-929 ; layout web/compute.wgsl:4-13
#include "../../include/rtcore.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

struct rt_mesh {
  std::vector<float> pos, nor, tex;
  std::vector<uint32_t> vidx, nidx, tidx;
  std::vector<int32_t> mat_ids;
  std::vector<rt_material> materials;
  std::vector<std::map<std::string, std::string>> unknown;   // per material: material_t::unknown_parameter
};

namespace {

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_blank(char c) { return c == ' ' || c == '\t'; }

// Number grammar and arithmetic of tiny_obj_loader.h:954-1122: digits are accumulated in
// double (m = m*10 + d; fraction digits add d * 10^-k from a table for k < 8), a decimal
// exponent E is applied as ldexp(m * 5^E, E).  Not correctly rounded — kept on purpose so
// parsed floats equal what the reference's loader would produce.
bool parse_number(const char* s, const char* end, double* out) {
  if (s >= end) return false;
  static const double frac_lut[8] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
  const char* c = s;
  bool neg = false, dot_first = false;
  if (*c == '+' || *c == '-') {
    neg = (*c == '-');
    ++c;
    if (c != end && *c == '.') dot_first = true;
  } else if (*c == '.') {
    dot_first = true;
  } else if (!is_digit(*c)) {
    return false;
  }
  double m = 0.0;
  int ndig = 0;
  if (!dot_first) {
    while (c != end && is_digit(*c)) { m *= 10; m += (int)(*c - '0'); ++c; ++ndig; }
    if (ndig == 0) return false;
  }
  int exp10 = 0;
  if (c != end && *c == '.') {
    ++c;
    int k = 1;
    while (c != end && is_digit(*c)) {
      m += (int)(*c - '0') * (k < 8 ? frac_lut[k] : std::pow(10.0, -k));
      ++k; ++c;
    }
  } else if (c != end && (*c == 'e' || *c == 'E')) {
    // handled below
  } else {
    *out = neg ? -m : m;
    return true;
  }
  if (c != end && (*c == 'e' || *c == 'E')) {
    ++c;
    bool eneg = false;
    if (c != end && (*c == '+' || *c == '-')) { eneg = (*c == '-'); ++c; }
    else if (c == end || !is_digit(*c)) return false;
    int nd = 0;
    while (c != end && is_digit(*c)) {
      if (exp10 > 2147483647 / 10) return false;
      exp10 = exp10 * 10 + (int)(*c - '0');
      ++c; ++nd;
    }
    if (nd == 0) return false;
    if (eneg) exp10 = -exp10;
  }
  double v = exp10 ? std::ldexp(m * std::pow(5.0, exp10), exp10) : m;
  *out = neg ? -v : v;
  return true;
}

struct Cursor {
  const char* p;
  void skip_blank() { while (is_blank(*p)) ++p; }
  const char* token_end() const { const char* e = p; while (*e && *e != ' ' && *e != '\t' && *e != '\r') ++e; return e; }
  float real(double dflt = 0.0) {                 // parseReal, tiny_obj_loader.h:1124-1133
    skip_blank();
    const char* e = token_end();
    double v = dflt;
    parse_number(p, e, &v);
    p = e;
    return (float)v;
  }
  std::string word() { skip_blank(); const char* e = token_end(); std::string s(p, e); p = e; return s; }
  std::string rest() {
    skip_blank();
    std::string s(p);
    while (!s.empty() && (s.back() == '\r' || s.back() == '\n' || is_blank(s.back()))) s.pop_back();
    return s;
  }
  bool at_end() { skip_blank(); return *p == 0 || *p == '\r' || *p == '\n'; }
};

struct Corner { int v, vt, vn; };

// index fix-up, tiny_obj_loader.h:867-907: 1-based -> 0-based, negative = relative to count
bool fix_index(int raw, int count, bool allow_zero, int* out) {
  if (raw > 0) { *out = raw - 1; return true; }
  if (raw == 0) { *out = -1; return allow_zero; }
  *out = count + raw;
  return *out >= 0;
}
inline void skip_to_sep(const char*& p) { while (*p && *p != '/' && *p != ' ' && *p != '\t' && *p != '\r') ++p; }

// i, i/j, i//k, i/j/k — tiny_obj_loader.h:1312-1372
bool parse_corner(const char*& p, int nv, int nvn, int nvt, Corner* c) {
  c->v = c->vt = c->vn = -1;
  if (!fix_index(atoi(p), nv, false, &c->v)) return false;
  skip_to_sep(p);
  if (*p != '/') return true;
  ++p;
  if (*p == '/') {
    ++p;
    if (!fix_index(atoi(p), nvn, true, &c->vn)) return false;
    skip_to_sep(p);
    return true;
  }
  if (!fix_index(atoi(p), nvt, true, &c->vt)) return false;
  skip_to_sep(p);
  if (*p != '/') return true;
  ++p;
  if (!fix_index(atoi(p), nvn, true, &c->vn)) return false;
  skip_to_sep(p);
  return true;
}

// ---- LoadMtl, tiny_obj_loader.h:2322-2773, restated branch by branch (same order of the key tests, same IS_SPACE guards) ----
inline bool is_eol(char c) { return c == '\r' || c == '\n' || c == 0; }                 // IS_NEW_LINE :850
inline bool key_is(const char* t, const char* key) {                                   // strncmp(token, key, n) == 0 && IS_SPACE(token[n])
  const size_t n = std::strlen(key);
  return std::strncmp(t, key, n) == 0 && is_blank(t[n]);
}
float mtl_real(const char*& t, double dflt = 0.0) {                                    // parseReal :1124-1133
  Cursor c{t};
  const float v = c.real(dflt);
  t = c.p;
  return v;
}
std::string mtl_word(const char*& t) {                                                 // parseString :909-917
  Cursor c{t};
  std::string s = c.word();
  t = c.p;
  return s;
}
int mtl_int(const char*& t) {                                                          // parseInt :919-925
  while (is_blank(*t)) ++t;
  const int i = atoi(t);
  while (*t && *t != ' ' && *t != '\t' && *t != '\r') ++t;
  return i;
}
bool mtl_on_off(const char*& t) {                                                      // parseOnOff :1224-1241, called with default_value = true
  while (is_blank(*t)) ++t;
  bool ret = true;
  if (std::strncmp(t, "on", 2) == 0) ret = true;
  else if (std::strncmp(t, "off", 3) == 0) ret = false;
  while (*t && *t != ' ' && *t != '\t' && *t != '\r') ++t;
  return ret;
}
int mtl_texture_type(const char*& t) {                                                 // parseTextureType :1243-1281 (prefix matches, in its order)
  while (is_blank(*t)) ++t;
  static const struct { const char* name; int ty; } types[] = {{"cube_top", 2}, {"cube_bottom", 3}, {"cube_left", 6}, {"cube_right", 7},
                                                               {"cube_front", 4}, {"cube_back", 5}, {"sphere", 1}};
  int ty = 0;
  for (const auto& k : types) if (std::strncmp(t, k.name, std::strlen(k.name)) == 0) { ty = k.ty; break; }
  while (*t && *t != ' ' && *t != '\t' && *t != '\r') ++t;
  return ty;
}
void init_texopt(rt_texopt* o, bool is_bump) {                                         // InitTexOpt :1530-1558
  std::memset(o, 0, sizeof(*o));
  o->imfchan = is_bump ? 'l' : 'm';
  o->bump_multiplier = 1.0f; o->clamp = 0; o->blendu = 1; o->blendv = 1; o->sharpness = 1.0f; o->brightness = 0.0f; o->contrast = 1.0f;
  for (int i = 0; i < 3; ++i) { o->origin_offset[i] = 0.0f; o->scale[i] = 1.0f; o->turbulence[i] = 0.0f; }
  o->texture_resolution = -1; o->type = 0;
}
// ParseTextureNameAndOption :1411-1528.  Options are recognised only IN FRONT of the file name; the name is everything from the
// first token that is no option to the end of the line (blanks and later "-options" included).  -> false: no name on the line
// (*texname is then left as it was).
bool parse_texture_name_and_option(std::string* texname, rt_texopt* o, const char* linebuf) {
  bool found = false;
  std::string name;
  const char* t = linebuf;
  while (!is_eol(*t)) {
    while (is_blank(*t)) ++t;
    if (key_is(t, "-blendu")) { t += 8; o->blendu = mtl_on_off(t) ? 1 : 0; }
    else if (key_is(t, "-blendv")) { t += 8; o->blendv = mtl_on_off(t) ? 1 : 0; }
    else if (key_is(t, "-clamp")) { t += 7; o->clamp = mtl_on_off(t) ? 1 : 0; }
    else if (key_is(t, "-boost")) { t += 7; o->sharpness = mtl_real(t, 1.0); }
    else if (key_is(t, "-bm")) { t += 4; o->bump_multiplier = mtl_real(t, 1.0); }
    else if (key_is(t, "-o")) { t += 3; for (int i = 0; i < 3; ++i) o->origin_offset[i] = mtl_real(t, 0.0); }
    else if (key_is(t, "-s")) { t += 3; for (int i = 0; i < 3; ++i) o->scale[i] = mtl_real(t, 1.0); }
    else if (key_is(t, "-t")) { t += 3; for (int i = 0; i < 3; ++i) o->turbulence[i] = mtl_real(t, 0.0); }
    else if (key_is(t, "-type")) { t += 5; o->type = mtl_texture_type(t); }
    else if (key_is(t, "-texres")) { t += 7; o->texture_resolution = mtl_int(t); }
    else if (key_is(t, "-imfchan")) {
      t += 9;
      while (is_blank(*t)) ++t;
      const char* e = t;
      while (*e && *e != ' ' && *e != '\t' && *e != '\r') ++e;
      if (e - t == 1) o->imfchan = *t;
      t = e;
    }
    else if (key_is(t, "-mm")) { t += 4; o->brightness = mtl_real(t, 0.0); o->contrast = mtl_real(t, 1.0); }
    else if (key_is(t, "-colorspace")) { t += 12; std::snprintf(o->colorspace, sizeof(o->colorspace), "%s", mtl_word(t).c_str()); }
    else { name = std::string(t); t += name.size(); found = true; }
  }
  if (found) *texname = name;
  return found;
}

void init_material(rt_material* m) {                                                   // InitMaterial :1560-1612
  std::memset(m, 0, sizeof(*m));
  for (int k = 0; k < RT_N_MAPS; ++k) init_texopt(&m->texopt[k], k == RT_MAP_BUMP);
  m->illum = 0; m->dissolve = 1.0f; m->shininess = 1.0f; m->ior = 1.0f;              // every colour, Kd included, starts at 0 (:1585-1592)
}

// materials / unknown: appended to; by_name: material_map (std::map::insert keeps the FIRST material of a name, :2390, :2765)
void parse_mtl(const std::string& text, std::vector<rt_material>& mats, std::vector<std::map<std::string, std::string>>& unknown,
               std::map<std::string, int>& by_name) {
  std::istringstream in(text);
  std::string line;
  rt_material cur;                       // "Create a default material anyway" (:2328): lines in front of the first newmtl land in it
  std::string cur_name, texname[RT_N_MAPS];
  std::map<std::string, std::string> cur_unknown;
  bool has_d = false, has_tr = false, has_kd = false;
  init_material(&cur);
  auto flush = [&]() {
    std::snprintf(cur.name, sizeof(cur.name), "%s", cur_name.c_str());
    for (int k = 0; k < RT_N_MAPS; ++k) std::snprintf(cur.texname[k], sizeof(cur.texname[k]), "%s", texname[k].c_str());
    std::snprintf(cur.diffuse_texname, sizeof(cur.diffuse_texname), "%s", texname[RT_MAP_KD].c_str());
    std::snprintf(cur.bump_texname, sizeof(cur.bump_texname), "%s", texname[RT_MAP_BUMP].c_str());
    by_name.insert(std::make_pair(cur_name, (int)mats.size()));
    mats.push_back(cur); unknown.push_back(cur_unknown);
  };
  auto real3 = [&](const char*& t, float* v) { for (int i = 0; i < 3; ++i) v[i] = mtl_real(t, 0.0); };
  auto tex = [&](int k, const char* t) { parse_texture_name_and_option(&texname[k], &cur.texopt[k], t); };
  while (std::getline(in, line)) {
    // trailing blanks, then one '\r', go first (:2349-2365)
    line.erase(line.find_last_not_of(" \t") + 1);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const char* t = line.c_str();
    while (is_blank(*t)) ++t;
    if (*t == 0 || *t == '#') continue;
    if (key_is(t, "newmtl")) {                                    // :2385-2417: the previous material is kept only if it has a name
      if (!cur_name.empty()) flush();
      init_material(&cur); cur_unknown.clear();
      for (auto& n : texname) n.clear();
      has_d = has_tr = has_kd = false;
      t += 7;
      cur_name = mtl_word(t);                                     // ONE word (parseString), not the rest of the line
      continue;
    }
    if (t[0] == 'K' && t[1] == 'a' && is_blank(t[2])) { t += 2; real3(t, cur.ambient); continue; }
    if (t[0] == 'K' && t[1] == 'd' && is_blank(t[2])) { t += 2; real3(t, cur.diffuse); has_kd = true; continue; }
    if (t[0] == 'K' && t[1] == 's' && is_blank(t[2])) { t += 2; real3(t, cur.specular); continue; }
    if ((t[0] == 'K' && t[1] == 't' && is_blank(t[2])) || (t[0] == 'T' && t[1] == 'f' && is_blank(t[2]))) { t += 2; real3(t, cur.transmittance); continue; }
    if (t[0] == 'N' && t[1] == 'i' && is_blank(t[2])) { t += 2; cur.ior = mtl_real(t); continue; }
    if (t[0] == 'K' && t[1] == 'e' && is_blank(t[2])) { t += 2; real3(t, cur.emission); continue; }
    if (t[0] == 'N' && t[1] == 's' && is_blank(t[2])) { t += 2; cur.shininess = mtl_real(t); continue; }
    if (key_is(t, "illum")) { t += 6; cur.illum = mtl_int(t); continue; }
    if (t[0] == 'd' && is_blank(t[1])) { t += 1; cur.dissolve = mtl_real(t); has_d = true; continue; }           // :2506-2520
    if (t[0] == 'T' && t[1] == 'r' && is_blank(t[2])) {                                                          // :2521-2541: `d` wins
      t += 2;
      if (!has_d) cur.dissolve = 1.0f - mtl_real(t);
      has_tr = true;
      continue;
    }
    if (t[0] == 'P' && t[1] == 'r' && is_blank(t[2])) { t += 2; cur.roughness = mtl_real(t); continue; }
    if (t[0] == 'P' && t[1] == 'm' && is_blank(t[2])) { t += 2; cur.metallic = mtl_real(t); continue; }
    if (t[0] == 'P' && t[1] == 's' && is_blank(t[2])) { t += 2; cur.sheen = mtl_real(t); continue; }
    if (t[0] == 'P' && t[1] == 'c' && is_blank(t[2])) { t += 2; cur.clearcoat_thickness = mtl_real(t); continue; }
    if (key_is(t, "Pcr")) { t += 4; cur.clearcoat_roughness = mtl_real(t); continue; }
    if (key_is(t, "aniso")) { t += 6; cur.anisotropy = mtl_real(t); continue; }
    if (key_is(t, "anisor")) { t += 7; cur.anisotropy_rotation = mtl_real(t); continue; }
    if (key_is(t, "map_Ka")) { tex(RT_MAP_KA, t + 7); continue; }
    if (key_is(t, "map_Kd")) {                                    // :2609-2625: 0.6 only when no Kd line came BEFORE this one
      tex(RT_MAP_KD, t + 7);
      if (!has_kd) cur.diffuse[0] = cur.diffuse[1] = cur.diffuse[2] = 0.6f;
      continue;
    }
    if (key_is(t, "map_Ks")) { tex(RT_MAP_KS, t + 7); continue; }
    if (key_is(t, "map_Ns")) { tex(RT_MAP_NS, t + 7); continue; }
    if (key_is(t, "map_bump") || key_is(t, "map_Bump")) { tex(RT_MAP_BUMP, t + 9); continue; }
    if (key_is(t, "bump")) { tex(RT_MAP_BUMP, t + 5); continue; }
    if (key_is(t, "map_d")) { t += 6; texname[RT_MAP_D] = t; tex(RT_MAP_D, t); continue; }   // :2666-2673: the raw rest first
    if (key_is(t, "map_disp") || key_is(t, "map_Disp")) { tex(RT_MAP_DISP, t + 9); continue; }
    if (key_is(t, "disp")) { tex(RT_MAP_DISP, t + 5); continue; }
    if (key_is(t, "refl")) { tex(RT_MAP_REFL, t + 5); continue; }
    if (key_is(t, "map_Pr")) { tex(RT_MAP_PR, t + 7); continue; }
    if (key_is(t, "map_Pm")) { tex(RT_MAP_PM, t + 7); continue; }
    if (key_is(t, "map_Ps")) { tex(RT_MAP_PS, t + 7); continue; }
    if (key_is(t, "map_Ke")) { tex(RT_MAP_KE, t + 7); continue; }
    if (key_is(t, "norm")) { tex(RT_MAP_NORM, t + 5); continue; }
    // :2749-2762: the key ends at the first ' ' of the line (at the first '\t' only when the line has no ' ' at all), the
    // value is everything behind that one character; map::insert keeps the first value of a key
    const char* sep = std::strchr(t, ' ');
    if (!sep) sep = std::strchr(t, '\t');
    if (sep) cur_unknown.insert(std::make_pair(std::string(t, (size_t)(sep - t)), std::string(sep + 1)));
  }
  (void)has_tr;
  flush();                                                        // :2764-2767: the last material always, named or not
}

// Point-in-polygon test of tiny_obj_loader.h:1616-1628 (W. R. Franklin's pnpoly) on a triangle, f32 as there.
bool point_in_tri(const float vx[3], const float vy[3], float tx, float ty) {
  bool in = false;
  for (int i = 0, j = 2; i < 3; j = i++)
    if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) in = !in;
  return in;
}

// tinyobj's built-in ear clipping of a polygon with more than 4 corners (tiny_obj_loader.h:1945-2191), f32 arithmetic:
//   * work plane: the first corner triple (k, k+1, k+2) whose edge cross product has a component above FLT_EPSILON picks
//     the two axes - |cx| largest: (y, z); |cz| largest: (x, y); otherwise (x, z); no such corner: (y, z);
//   * a candidate ear (g, g+1, g+2) of the remaining ring is skipped when cross * area < 0, where cross is the 2-D cross
//     product of its two edges and "area" is HALF THE CROSS PRODUCT OF THE POSITION VECTORS of its first two corners (so
//     the result depends on where the polygon sits relative to the origin - restated as it is), or when any other
//     remaining corner lies inside it (point_in_tri);
//   * an ear emits (g, g+1, g+2) and drops g+1; g does not advance; the search gives up after a full round without
//     progress (the corners still left are dropped, as there); the last three corners are emitted as they stand.
// P: positions (3 floats per vertex); vidx[i]: position index of corner i; tris: corner numbers, 3 per triangle.
void ear_clip(const float* P, const std::vector<int>& vidx, std::vector<int>& tris) {
  const size_t n = vidx.size();
  size_t axes[2] = {1, 2};
  for (size_t k = 0; k < n; ++k) {
    const float* a = P + 3 * (size_t)vidx[k % n];
    const float* b = P + 3 * (size_t)vidx[(k + 1) % n];
    const float* c = P + 3 * (size_t)vidx[(k + 2) % n];
    const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
    const float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
    const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
    const float eps = 1.1920928955078125e-07f;   // std::numeric_limits<float>::epsilon()
    if (cx > eps || cy > eps || cz > eps) {
      if (!(cx > cy && cx > cz)) {
        axes[0] = 0;
        if (cz > cx && cz > cy) axes[1] = 1;
      }
      break;
    }
  }
  std::vector<int> ring(n);                        // corner numbers still in the polygon
  for (size_t i = 0; i < n; ++i) ring[i] = (int)i;
  size_t guess = 0, budget = n, prev = n;
  while (ring.size() > 3 && budget > 0) {
    const size_t m = ring.size();
    if (guess >= m) guess -= m;
    if (prev != m) { prev = m; budget = m; } else budget--;
    int corner[3];
    float vx[3], vy[3];
    for (int k = 0; k < 3; ++k) {
      corner[k] = ring[(guess + k) % m];
      vx[k] = P[3 * (size_t)vidx[corner[k]] + axes[0]];
      vy[k] = P[3 * (size_t)vidx[corner[k]] + axes[1]];
    }
    const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
    const float cross = e0x * e1y - e0y * e1x;
    const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
    if (cross * area < 0.0f) { guess += 1; continue; }
    bool overlap = false;
    for (size_t other = 3; other < m && !overlap; ++other) {
      const int o = ring[(guess + other) % m];
      overlap = point_in_tri(vx, vy, P[3 * (size_t)vidx[o] + axes[0]], P[3 * (size_t)vidx[o] + axes[1]]);
    }
    if (overlap) { guess += 1; continue; }
    tris.push_back(corner[0]); tris.push_back(corner[1]); tris.push_back(corner[2]);
    ring.erase(ring.begin() + (ptrdiff_t)((guess + 1) % m));
  }
  if (ring.size() == 3) { tris.push_back(ring[0]); tris.push_back(ring[1]); tris.push_back(ring[2]); }
}

void set_err(char* err, size_t n, const std::string& msg) { if (err && n) std::snprintf(err, n, "%s", msg.c_str()); }

int parse_obj(const std::string& text, const std::string* mtl_text, const std::string& mtl_dir,
              rt_mesh** out, char* err, size_t err_len) {
  auto* m = new rt_mesh();
  std::map<std::string, int> mat_by_name;
  std::set<std::string> mtl_loaded;
  if (mtl_text) parse_mtl(*mtl_text, m->materials, m->unknown, mat_by_name);
  std::vector<Corner> face;
  int cur_mat = -1;
  size_t line_no = 0;
  bool any_vn = false, any_vt = false;
  std::vector<Corner> corners;  // 3 per output triangle
  const char* p = text.c_str();
  const char* text_end = p + text.size();
  std::string line;
  while (p < text_end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(text_end - p));
    const char* le = nl ? nl : text_end;
    line.assign(p, le);
    p = nl ? nl + 1 : text_end;
    ++line_no;
    Cursor c{line.c_str()};
    c.skip_blank();
    const char* t = c.p;
    if (t[0] == 0 || t[0] == '#' || t[0] == '\r') continue;
    if (t[0] == 'v' && is_blank(t[1])) {
      c.p = t + 2;
      float x = c.real(), y = c.real(), z = c.real();
      m->pos.push_back(x); m->pos.push_back(y); m->pos.push_back(z);
    } else if (t[0] == 'v' && t[1] == 'n' && is_blank(t[2])) {
      c.p = t + 3;
      float x = c.real(), y = c.real(), z = c.real();
      m->nor.push_back(x); m->nor.push_back(y); m->nor.push_back(z);
      any_vn = true;
    } else if (t[0] == 'v' && t[1] == 't' && is_blank(t[2])) {
      c.p = t + 3;
      float u = c.real(), v = c.real();
      m->tex.push_back(u); m->tex.push_back(v);
      any_vt = true;
    } else if (t[0] == 'f' && is_blank(t[1])) {
      c.p = t + 2;
      face.clear();
      const int nv = (int)(m->pos.size() / 3), nvn = (int)(m->nor.size() / 3), nvt = (int)(m->tex.size() / 2);
      while (!c.at_end()) {
        Corner k;
        if (!parse_corner(c.p, nv, nvn, nvt, &k)) {
          set_err(err, err_len, "invalid face index at line " + std::to_string(line_no));
          delete m; return RT_ERR_IO;
        }
        if (k.v >= nv || k.vn >= nvn || k.vt >= nvt) {
          set_err(err, err_len, "face index out of range at line " + std::to_string(line_no));
          delete m; return RT_ERR_IO;
        }
        face.push_back(k);
      }
      const size_t n = face.size();
      if (n < 3) continue;  // degenerate face: skipped with a warning in tiny_obj_loader
      auto emit = [&](int a, int b, int cc) {
        corners.push_back(face[a]); corners.push_back(face[b]); corners.push_back(face[cc]);
        m->mat_ids.push_back(cur_mat);
      };
      if (n == 3) emit(0, 1, 2);
      else if (n == 4) {
        // split along the shorter diagonal, f32 arithmetic (tiny_obj_loader.h:1756-1802)
        const float* P = m->pos.data();
        auto V = [&](int i, int k) { return P[3 * face[i].v + k]; };
        float e02[3], e13[3];
        for (int k = 0; k < 3; ++k) { e02[k] = V(2, k) - V(0, k); e13[k] = V(3, k) - V(1, k); }
        float s02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
        float s13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
        if (s02 < s13) { emit(0, 1, 2); emit(0, 2, 3); }
        else { emit(0, 1, 3); emit(1, 2, 3); }
      } else {
        std::vector<int> vi(n), tris;
        for (size_t i = 0; i < n; ++i) vi[i] = face[i].v;
        ear_clip(m->pos.data(), vi, tris);
        for (size_t i = 0; i + 2 < tris.size(); i += 3) emit(tris[i], tris[i + 1], tris[i + 2]);
      }
    } else if (!strncmp(t, "usemtl", 6)) {              // tiny_obj_loader.h:3200-3233: no blank needed behind the key; ONE word
      const char* q = t + 6;
      const std::string name = mtl_word(q);
      auto it = mat_by_name.find(name);
      cur_mat = (it == mat_by_name.end()) ? -1 : it->second;   // unknown name: a warning there, material id -1
    } else if (!strncmp(t, "mtllib", 6) && is_blank(t[6])) {
      // :3236-3303: the rest of the line split at ' ' ('\' escapes a character; tabs do not split); the files are tried in
      // order, the first one that opens is loaded and ends the search; a file loaded before is not loaded again
      if (!mtl_text && !mtl_dir.empty()) {
        std::string rest(t + 7);
        while (!rest.empty() && (rest.back() == '\r' || rest.back() == '\n')) rest.pop_back();
        std::vector<std::string> names;
        std::string tok;
        bool esc = false;
        for (char ch : rest) {
          if (esc) esc = false;
          else if (ch == '\\') { esc = true; continue; }
          else if (ch == ' ') { if (!tok.empty()) names.push_back(tok); tok.clear(); continue; }
          tok += ch;
        }
        names.push_back(tok);
        for (const std::string& fn : names) {
          if (mtl_loaded.count(fn)) continue;
          std::ifstream f(mtl_dir + "/" + fn, std::ios::binary);
          if (!f) continue;                                     // a missing .mtl is a warning, not an error
          std::stringstream ss; ss << f.rdbuf();
          parse_mtl(ss.str(), m->materials, m->unknown, mat_by_name);
          mtl_loaded.insert(fn);
          break;
        }
      }
    }
    // g / o / s / anything else: no effect on the flattened triangle list
  }
  // dummy entries so that "no normal" / "no texcoord" corners still index valid memory
  const uint32_t dummy_n = (uint32_t)(m->nor.size() / 3), dummy_t = (uint32_t)(m->tex.size() / 2);
  m->nor.insert(m->nor.end(), {0.0f, 0.0f, 0.0f});
  m->tex.insert(m->tex.end(), {0.0f, 0.0f});
  (void)any_vn; (void)any_vt;
  m->vidx.reserve(corners.size()); m->nidx.reserve(corners.size()); m->tidx.reserve(corners.size());
  for (const Corner& k : corners) {
    m->vidx.push_back((uint32_t)k.v);
    m->nidx.push_back(k.vn < 0 ? dummy_n : (uint32_t)k.vn);
    m->tidx.push_back(k.vt < 0 ? dummy_t : (uint32_t)k.vt);
  }
  *out = m;
  return RT_OK;
}

}  // namespace

extern "C" int rt_mesh_parse_obj(const char* obj_text, size_t obj_len, const char* mtl_text, size_t mtl_len,
                                 rt_mesh** out, char* err, size_t err_len) {
  if (!obj_text || !out) return RT_ERR_INVALID;
  std::string obj(obj_text, obj_len), mtl;
  if (mtl_text) mtl.assign(mtl_text, mtl_len);
  return parse_obj(obj, mtl_text ? &mtl : nullptr, "", out, err, err_len);
}

extern "C" int rt_mesh_load_obj(const char* obj_path, const char* mtl_dir, rt_mesh** out, char* err, size_t err_len) {
  if (!obj_path || !out) return RT_ERR_INVALID;
  std::ifstream f(obj_path, std::ios::binary);
  if (!f) { set_err(err, err_len, std::string("cannot open ") + obj_path); return RT_ERR_IO; }
  std::stringstream ss; ss << f.rdbuf();
  std::string dir;
  if (mtl_dir) dir = mtl_dir;
  else {
    std::string pth(obj_path);
    size_t k = pth.find_last_of('/');
    dir = (k == std::string::npos) ? "." : pth.substr(0, k);
  }
  return parse_obj(ss.str(), nullptr, dir, out, err, err_len);
}

extern "C" int rt_mesh_view_get(const rt_mesh* m, rt_mesh_view* v) {
  if (!m || !v) return RT_ERR_INVALID;
  v->positions = m->pos.data(); v->n_vert = (uint32_t)(m->pos.size() / 3);
  v->normals = m->nor.data(); v->n_nor = (uint32_t)(m->nor.size() / 3);
  v->texcoords = m->tex.data(); v->n_tex = (uint32_t)(m->tex.size() / 2);
  v->vert_idx = m->vidx.data(); v->nor_idx = m->nidx.data(); v->tex_idx = m->tidx.data();
  v->material_ids = m->mat_ids.data();
  v->n_tri = (uint32_t)(m->vidx.size() / 3);
  v->n_materials = (uint32_t)m->materials.size();
  return RT_OK;
}
extern "C" int rt_mesh_material(const rt_mesh* m, uint32_t i, rt_material* out) {
  if (!m || !out || i >= m->materials.size()) return RT_ERR_INVALID;
  *out = m->materials[i];
  return RT_OK;
}
extern "C" uint32_t rt_mesh_material_n_unknown(const rt_mesh* m, uint32_t i) {
  return (m && i < m->unknown.size()) ? (uint32_t)m->unknown[i].size() : 0u;
}
extern "C" int rt_mesh_material_unknown(const rt_mesh* m, uint32_t i, uint32_t k, const char** key, const char** value) {
  if (!m || !key || !value || i >= m->unknown.size() || k >= m->unknown[i].size()) return RT_ERR_INVALID;
  auto it = m->unknown[i].begin();
  std::advance(it, k);
  *key = it->first.c_str(); *value = it->second.c_str();
  return RT_OK;
}
extern "C" void rt_mesh_free(rt_mesh* m) { delete m; }

// web/script.js:207-227 — bbox over all vertices, centre, scale by 1/max-extent; JS numbers
// are doubles, the Float32Array store rounds to f32.
extern "C" int rt_normalize_mesh(float* p, uint32_t n_vert) {
  if (!p) return RT_ERR_INVALID;
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  for (uint32_t i = 0; i < n_vert; ++i)
    for (int k = 0; k < 3; ++k) {
      const double v = p[3 * (size_t)i + k];
      if (v < lo[k]) lo[k] = v;
      if (v > hi[k]) hi[k] = v;
    }
  double mid[3], span = 0.0;
  for (int k = 0; k < 3; ++k) { mid[k] = (lo[k] + hi[k]) * 0.5; span = std::fmax(span, hi[k] - lo[k]); }
  if (n_vert == 0) return RT_OK;
  span = std::fmax(std::fmax(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  const double sc = 1 / span;
  for (size_t i = 0; i < (size_t)n_vert * 3; ++i) p[i] = (float)((p[i] - mid[i % 3]) * sc);
  return RT_OK;
}

// ------------------------------------------------------------------ uniforms
namespace {
struct M4 { float f[16]; };   // flat, as the Float32Array of script.js:57-74

M4 trs_matrix(const double r[3], const double s[3], const double t[3]) {   // script.js:30-75
  const double cx = std::cos(r[0]), sx = std::sin(r[0]);
  const double cy = std::cos(r[1]), sy = std::sin(r[1]);
  const double cz = std::cos(r[2]), sz = std::sin(r[2]);
  const double rot[3][3] = {                         // ZYX order
      {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx},
      {sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx},
      {-sy, cy * sx, cy * cx}};
  M4 m;
  for (int row = 0; row < 3; ++row) {
    for (int col = 0; col < 3; ++col) m.f[4 * row + col] = (float)(rot[row][col] * s[col]);
    m.f[4 * row + 3] = 0.0f;
  }
  m.f[12] = (float)t[0]; m.f[13] = (float)t[1]; m.f[14] = (float)t[2]; m.f[15] = 1.0f;
  return m;
}

M4 inverse(const M4& in) {                            // script.js:77-133 (cofactor expansion in double)
  double a[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a[i][j] = in.f[4 * i + j];
  const double b00 = a[0][0] * a[1][1] - a[0][1] * a[1][0], b01 = a[0][0] * a[1][2] - a[0][2] * a[1][0];
  const double b02 = a[0][0] * a[1][3] - a[0][3] * a[1][0], b03 = a[0][1] * a[1][2] - a[0][2] * a[1][1];
  const double b04 = a[0][1] * a[1][3] - a[0][3] * a[1][1], b05 = a[0][2] * a[1][3] - a[0][3] * a[1][2];
  const double b06 = a[2][0] * a[3][1] - a[2][1] * a[3][0], b07 = a[2][0] * a[3][2] - a[2][2] * a[3][0];
  const double b08 = a[2][0] * a[3][3] - a[2][3] * a[3][0], b09 = a[2][1] * a[3][2] - a[2][2] * a[3][1];
  const double b10 = a[2][1] * a[3][3] - a[2][3] * a[3][1], b11 = a[2][2] * a[3][3] - a[2][3] * a[3][2];
  double det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
  M4 o;
  for (float& v : o.f) v = 0.0f;                     // `new Float32Array(16)` stays zero when !det
  if (!(det != 0.0)) return o;
  det = 1.0 / det;
  const double r[16] = {
      (a[1][1] * b11 - a[1][2] * b10 + a[1][3] * b09) * det,  (-a[0][1] * b11 + a[0][2] * b10 - a[0][3] * b09) * det,
      (a[3][1] * b05 - a[3][2] * b04 + a[3][3] * b03) * det,  (-a[2][1] * b05 + a[2][2] * b04 - a[2][3] * b03) * det,
      (-a[1][0] * b11 + a[1][2] * b08 - a[1][3] * b07) * det, (a[0][0] * b11 - a[0][2] * b08 + a[0][3] * b07) * det,
      (-a[3][0] * b05 + a[3][2] * b02 - a[3][3] * b01) * det, (a[2][0] * b05 - a[2][2] * b02 + a[2][3] * b01) * det,
      (a[1][0] * b10 - a[1][1] * b08 + a[1][3] * b06) * det,  (-a[0][0] * b10 + a[0][1] * b08 - a[0][3] * b06) * det,
      (a[3][0] * b04 - a[3][1] * b02 + a[3][3] * b00) * det,  (-a[2][0] * b04 + a[2][1] * b02 - a[2][3] * b00) * det,
      (-a[1][0] * b09 + a[1][1] * b07 - a[1][2] * b06) * det, (a[0][0] * b09 - a[0][1] * b07 + a[0][2] * b06) * det,
      (-a[3][0] * b03 + a[3][1] * b01 - a[3][2] * b00) * det, (a[2][0] * b03 - a[2][1] * b01 + a[2][2] * b00) * det};
  for (int i = 0; i < 16; ++i) o.f[i] = (float)r[i];
  return o;
}

struct D3 { double x, y, z; };
D3 unit(D3 v) {                                      // script.js:678-681
  double l = std::hypot(v.x, v.y, v.z);
  if (!(l != 0.0)) l = 1.0;
  return D3{v.x / l, v.y / l, v.z / l};
}
D3 crossd(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
}  // namespace

extern "C" void rt_camera_default(rt_camera* cam, uint32_t width, uint32_t height) {
  if (!cam) return;
  std::memset(cam, 0, sizeof(*cam));
  cam->pos[2] = 5.0;                                  // script.js:4
  for (int i = 0; i < 3; ++i) cam->scale[i] = 1.0;   // web/index.html:12-22 slider defaults
  cam->width = width; cam->height = height;
}

extern "C" int rt_make_uniforms(const rt_camera* cam, float u[44]) {
  if (!cam || !u || cam->width == 0 || cam->height == 0) return RT_ERR_INVALID;
  const double pi = 3.141592653589793;               // Math.PI
  const double cp = std::cos(cam->pitch), sp = std::sin(cam->pitch);
  const double cy = std::cos(cam->yaw), sy = std::sin(cam->yaw);
  const D3 look{cp * sy, sp, -cp * cy};              // script.js:596
  const D3 eye{cam->pos[0], cam->pos[1], cam->pos[2]};
  const D3 tgt{eye.x + look.x, eye.y + look.y, eye.z + look.z};
  const D3 fwd = unit(D3{tgt.x - eye.x, tgt.y - eye.y, tgt.z - eye.z});
  const D3 right = unit(crossd(fwd, D3{0, 1, 0}));
  const D3 up = crossd(right, fwd);
  const double fov = pi / 3, aspect = (double)cam->width / (double)cam->height, fl = 1 / std::tan(fov / 2);
  const double rot[3] = {(cam->rot_deg[0] * pi) / 180 - pi, (cam->rot_deg[1] * pi) / 180, (cam->rot_deg[2] * pi) / 180};
  const M4 inv = inverse(trs_matrix(rot, cam->scale, cam->translate));
  for (int i = 0; i < 44; ++i) u[i] = 0.0f;
  u[0] = (float)eye.x; u[1] = (float)eye.y; u[2] = (float)eye.z;
  u[4] = 5.0f; u[5] = -5.0f; u[6] = 5.0f; u[7] = 0.3f;                 // script.js:904-907
  u[8] = (float)(right.x * fl * aspect); u[9] = (float)(right.y * fl * aspect); u[10] = (float)(right.z * fl * aspect);
  u[12] = (float)(up.x * fl); u[13] = (float)(up.y * fl); u[14] = (float)(up.z * fl);
  u[16] = (float)(fwd.x * fl); u[17] = (float)(fwd.y * fl); u[18] = (float)(fwd.z * fl);
  for (int i = 0; i < 16; ++i) u[20 + i] = inv.f[i];
  u[36] = (float)cam->width; u[37] = (float)cam->height;
  u[38] = cam->time;
  return RT_OK;
}
