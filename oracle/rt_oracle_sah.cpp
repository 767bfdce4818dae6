/*
 * rt_oracle_sah.cpp — CPU ORACLE restatement of the reference's actual BVH builder.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Restates, statement by statement, web/script.js:285-513 (`buildLBVH`: despite its name a
 * recursive binned-SAH builder over centroids, 20 bins, median fallback) and the node part of
 * `flattenForSSBO` (web/script.js:546-559).  JS numbers are IEEE doubles; Float32Array stores
 * round to f32 — the same points round here.  Math.min / Math.max follow ECMA-262: NaN if any
 * operand is NaN, and -0 < +0.
 *
 * The recursion is kept as a recursion (it is the reference's control flow and fixes the DFS
 * pre-order numbering `left child = parent + 1`, script.js:331-332); it runs on a thread with a
 * large stack because SAH trees over degenerate inputs get deep.
 */
#include "rt_oracle.h"

#include <pthread.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline double js_min(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == b) return std::signbit(a) ? a : b;   // Math.min(+0, -0) = -0
  return a < b ? a : b;
}
inline double js_max(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == b) return std::signbit(a) ? b : a;   // Math.max(+0, -0) = +0
  return a > b ? a : b;
}

struct JsNode { double mn[3], mx[3]; double left, right; };

struct SahBuild {
  const float* positions; const uint32_t* vertIdx; uint32_t tris; int leaf;
  std::vector<float> triMin, triMax, centroids;
  std::vector<uint32_t> triOrder;
  std::vector<JsNode> nodes;
  int error = 0;

  // script.js:330-509
  int64_t recurse(int64_t start, int64_t end) {
    if (error) return -1;
    const int64_t nodeIdx = (int64_t)nodes.size();
    nodes.push_back(JsNode());
    const int64_t count = end - start;

    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = start; i < end; ++i) {
      const uint32_t t = triOrder[i];
      for (int k = 0; k < 3; ++k) {
        mn[k] = js_min(mn[k], triMin[3 * (size_t)t + k]);
        mx[k] = js_max(mx[k], triMax[3 * (size_t)t + k]);
      }
    }
    auto put = [&](double l, double r) {
      JsNode& n = nodes[nodeIdx];
      for (int k = 0; k < 3; ++k) { n.mn[k] = mn[k]; n.mx[k] = mx[k]; }
      n.left = l; n.right = r;
    };
    if (count <= leaf) { put((double)start, -(double)count); return nodeIdx; }

    double cMn[3] = {INFINITY, INFINITY, INFINITY}, cMx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = start; i < end; ++i) {
      const size_t off = 3 * (size_t)triOrder[i];
      for (int k = 0; k < 3; ++k) {
        const double v = centroids[off + k];
        cMn[k] = js_min(cMn[k], v);
        cMx[k] = js_max(cMx[k], v);
      }
    }
    const double ext[3] = {cMx[0] - cMn[0], cMx[1] - cMn[1], cMx[2] - cMn[2]};
    const int axis = ext[0] > ext[1] ? (ext[0] > ext[2] ? 0 : 2) : (ext[1] > ext[2] ? 1 : 2);

    const int BINS = 20;
    double binCount[BINS];
    double binMin[BINS][3], binMax[BINS][3];
    for (int b = 0; b < BINS; ++b) {
      binCount[b] = 0;
      for (int k = 0; k < 3; ++k) { binMin[b][k] = INFINITY; binMax[b][k] = -INFINITY; }
    }
    const double invExt = ext[axis] > 0 ? 1.0 / ext[axis] : 0;
    for (int64_t i = start; i < end; ++i) {
      const uint32_t t = triOrder[i];
      const double f = ((double)centroids[3 * (size_t)t + axis] - cMn[axis]) * invExt;
      double bf = std::floor(f * BINS);
      int b;
      if (bf < 0) b = 0;
      else if (bf >= BINS) b = BINS - 1;
      else if (bf != bf) { error = 2; return -1; }   // NaN bin index: `binCount[NaN]++` in JS; meshes with NaN are out of contract
      else b = (int)bf;
      binCount[b] += 1;
      for (int k = 0; k < 3; ++k) {
        const double v = centroids[3 * (size_t)t + k];
        binMin[b][k] = js_min(binMin[b][k], v);
        binMax[b][k] = js_max(binMax[b][k], v);
      }
    }

    double leftCount[BINS], leftMin3[BINS][3], leftMax3[BINS][3];
    double accCnt = 0, accMin[3] = {INFINITY, INFINITY, INFINITY}, accMax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < BINS; ++i) {
      accCnt += binCount[i];
      leftCount[i] = accCnt;
      for (int k = 0; k < 3; ++k) {
        accMin[k] = js_min(accMin[k], binMin[i][k]);
        accMax[k] = js_max(accMax[k], binMax[i][k]);
        leftMin3[i][k] = accMin[k];
        leftMax3[i][k] = accMax[k];
      }
    }
    double rightCount[BINS], rightMin3[BINS][3], rightMax3[BINS][3];
    accCnt = 0;
    for (int k = 0; k < 3; ++k) { accMin[k] = INFINITY; accMax[k] = -INFINITY; }
    for (int i = BINS - 1; i >= 0; --i) {
      accCnt += binCount[i];
      rightCount[i] = accCnt;
      for (int k = 0; k < 3; ++k) {
        accMin[k] = js_min(accMin[k], binMin[i][k]);
        accMax[k] = js_max(accMax[k], binMax[i][k]);
        rightMin3[i][k] = accMin[k];
        rightMax3[i][k] = accMax[k];
      }
    }

    const double PS = 2 * ((mx[0] - mn[0]) * (mx[1] - mn[1]) + (mx[1] - mn[1]) * (mx[2] - mn[2]) + (mx[2] - mn[2]) * (mx[0] - mn[0]));

    double bestCost = INFINITY;
    int bestBin = -1;
    for (int i = 0; i < BINS - 1; ++i) {
      const double nL = leftCount[i], nR = rightCount[i + 1];
      if (nL == 0 || nR == 0) continue;
      const double dL[3] = {leftMax3[i][0] - leftMin3[i][0], leftMax3[i][1] - leftMin3[i][1], leftMax3[i][2] - leftMin3[i][2]};
      const double dR[3] = {rightMax3[i + 1][0] - rightMin3[i + 1][0], rightMax3[i + 1][1] - rightMin3[i + 1][1],
                            rightMax3[i + 1][2] - rightMin3[i + 1][2]};
      const double sL = 2 * (dL[0] * dL[1] + dL[1] * dL[2] + dL[2] * dL[0]);
      const double sR = 2 * (dR[0] * dR[1] + dR[1] * dR[2] + dR[2] * dR[0]);
      const double cost = (sL / PS) * nL + (sR / PS) * nR;
      if (cost < bestCost) { bestCost = cost; bestBin = i; }
    }

    if (bestBin < 0) {
      const int64_t mid = start + (count >> 1);
      const int64_t l = recurse(start, mid);
      const int64_t r = recurse(mid, end);
      put((double)l, (double)r);
      return nodeIdx;
    }

    const double splitPos = cMn[axis] + ((double)(bestBin + 1) / BINS) * ext[axis];
    int64_t i = start, j = end - 1;
    while (i <= j) {
      const uint32_t t = triOrder[i];
      if ((double)centroids[3 * (size_t)t + axis] < splitPos) {
        i++;
      } else {
        const uint32_t tmp = triOrder[i]; triOrder[i] = triOrder[j]; triOrder[j] = tmp;
        j--;
      }
    }
    const int64_t mid = i;
    // The reference would recurse forever here (same range again -> RangeError in the browser):
    // report instead of reproducing a crash.
    if (mid == start || mid == end) { error = 1; return -1; }

    const int64_t leftIdx = recurse(start, mid);
    const int64_t rightIdx = recurse(mid, end);
    put((double)leftIdx, (double)rightIdx);
    return nodeIdx;
  }
};

void* sah_thread(void* arg) {
  SahBuild* b = static_cast<SahBuild*>(arg);
  b->recurse(0, (int64_t)b->tris);
  return nullptr;
}

}  // namespace

extern "C" int orc_build_sah(const float* positions, const float* normals, const uint32_t* vertIdx, const uint32_t* normIdx,
                             uint32_t n_tri, int leaf, float* nodes_out, uint32_t nodes_capacity, uint32_t* n_nodes_out,
                             float* tri_pos, float* tri_nor, uint32_t* order_out, double* seconds) {
  auto t0 = std::chrono::steady_clock::now();
  if (!positions || !vertIdx || n_tri == 0 || leaf < 1) return -1;
  SahBuild B;
  B.positions = positions; B.vertIdx = vertIdx; B.tris = n_tri; B.leaf = leaf;
  B.triMin.resize(3 * (size_t)n_tri); B.triMax.resize(3 * (size_t)n_tri); B.centroids.resize(3 * (size_t)n_tri);
  // script.js:293-322
  for (uint32_t t = 0; t < n_tri; ++t) {
    const size_t i0 = (size_t)vertIdx[3 * (size_t)t + 0] * 3, i1 = (size_t)vertIdx[3 * (size_t)t + 1] * 3, i2 = (size_t)vertIdx[3 * (size_t)t + 2] * 3;
    const double x0 = positions[i0], y0 = positions[i0 + 1], z0 = positions[i0 + 2];
    const double x1 = positions[i1], y1 = positions[i1 + 1], z1 = positions[i1 + 2];
    const double x2 = positions[i2], y2 = positions[i2 + 1], z2 = positions[i2 + 2];
    B.triMin[3 * (size_t)t + 0] = (float)js_min(js_min(x0, x1), x2);
    B.triMin[3 * (size_t)t + 1] = (float)js_min(js_min(y0, y1), y2);
    B.triMin[3 * (size_t)t + 2] = (float)js_min(js_min(z0, z1), z2);
    B.triMax[3 * (size_t)t + 0] = (float)js_max(js_max(x0, x1), x2);
    B.triMax[3 * (size_t)t + 1] = (float)js_max(js_max(y0, y1), y2);
    B.triMax[3 * (size_t)t + 2] = (float)js_max(js_max(z0, z1), z2);
    B.centroids[3 * (size_t)t + 0] = (float)((x0 + x1 + x2) / 3);
    B.centroids[3 * (size_t)t + 1] = (float)((y0 + y1 + y2) / 3);
    B.centroids[3 * (size_t)t + 2] = (float)((z0 + z1 + z2) / 3);
  }
  B.triOrder.resize(n_tri);
  for (uint32_t i = 0; i < n_tri; ++i) B.triOrder[i] = i;
  B.nodes.reserve(2 * (size_t)n_tri);

  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, (size_t)2 << 30);
  pthread_t th;
  if (pthread_create(&th, &attr, sah_thread, &B) != 0) { pthread_attr_destroy(&attr); return -4; }
  pthread_join(th, nullptr);
  pthread_attr_destroy(&attr);
  if (B.error) return -10 - B.error;

  if (n_nodes_out) *n_nodes_out = (uint32_t)B.nodes.size();
  if (B.nodes.size() > nodes_capacity) return -2;
  // flattenForSSBO, script.js:546-559 (Float32Array stores round the JS numbers to f32)
  for (size_t i = 0; i < B.nodes.size(); ++i) {
    const JsNode& n = B.nodes[i];
    float* b = nodes_out + 8 * i;
    b[0] = (float)n.mn[0]; b[1] = (float)n.mn[1]; b[2] = (float)n.mn[2]; b[3] = (float)n.left;
    b[4] = (float)n.mx[0]; b[5] = (float)n.mx[1]; b[6] = (float)n.mx[2]; b[7] = (float)n.right;
  }
  // script.js:527-545
  for (uint32_t i = 0; i < n_tri; ++i) {
    const uint32_t ti = B.triOrder[i];
    for (int v = 0; v < 3; ++v) {
      const size_t vi = vertIdx[3 * (size_t)ti + v];
      if (tri_pos) {
        float* d = tri_pos + ((size_t)i * 3 + v) * 4;
        d[0] = positions[3 * vi + 0]; d[1] = positions[3 * vi + 1]; d[2] = positions[3 * vi + 2]; d[3] = 0.0f;
      }
      if (tri_nor && normals && normIdx) {
        const size_t ni = normIdx[3 * (size_t)ti + v];
        float* d = tri_nor + ((size_t)i * 3 + v) * 4;
        d[0] = normals[3 * ni + 0]; d[1] = normals[3 * ni + 1]; d[2] = normals[3 * ni + 2]; d[3] = 0.0f;
      }
    }
  }
  if (order_out) std::memcpy(order_out, B.triOrder.data(), sizeof(uint32_t) * n_tri);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}
