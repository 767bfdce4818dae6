// ubench_stream.hip — what does the chip give a kernel with k_refit_local's memory shape at 1.31 M triangles?
// Per thread: one 128-byte record read (linear, or through a random permutation like the sort order), then W float4 stores
// into 48- / 32- / 64-byte-stride arrays (tri_pos, tri_nor, tri_pre, leaf node, internal node, wide record).
//   variant 0: stores only          variant 1: linear record read + stores      variant 2: permuted record read + stores
//   variant 3: permuted read only
//   variant 4 / 5: as 0 / 2, but every store instruction of a wave writes one contiguous 1 KB run (what staging the records
//   through LDS would give)
// Prints time and GB/s per variant at n = 1 310 720 and 16 x that (tail effects vs steady state).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ rec, const uint32_t* __restrict__ order, uint32_t n,
                                         float4* __restrict__ a48, float4* __restrict__ b48, float4* __restrict__ c48,
                                         float4* __restrict__ leaf, float4* __restrict__ node, float4* __restrict__ wide, float* sink) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float4 r[5];
  if (V == 0 || V == 4) { for (int i = 0; i < 5; ++i) r[i] = make_float4((float)j, 1.f, 2.f, (float)i); }
  else {
    const uint32_t ti = (V == 1) ? j : order[j];
    const float4* p = rec + 8 * (size_t)ti;
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i] = p[i];
  }
  if (V == 3) { if (r[0].x + r[1].y + r[2].z + r[3].w + r[4].x == 123.456f) sink[j] = 1.f; return; }
  if (V >= 4) {
    const uint32_t wbase = j & ~63u, l = j & 63u;       // the wave's 64 triangles: 3 KB / 2 KB / 4 KB contiguous per array
#pragma unroll
    for (int i = 0; i < 3; ++i) { a48[3 * (size_t)wbase + 64 * i + l] = r[i]; b48[3 * (size_t)wbase + 64 * i + l] = r[(i + 2) % 5]; c48[3 * (size_t)wbase + 64 * i + l] = r[(i + 1) % 5]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { leaf[2 * (size_t)wbase + 64 * i + l] = r[i]; node[2 * (size_t)wbase + 64 * i + l] = r[2 + i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) wide[4 * (size_t)wbase + 64 * i + l] = r[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { a48[3 * (size_t)j + i] = r[i]; b48[3 * (size_t)j + i] = r[(i + 2) % 5]; c48[3 * (size_t)j + i] = r[(i + 1) % 5]; }
  leaf[2 * (size_t)j] = r[0]; leaf[2 * (size_t)j + 1] = r[1];
  node[2 * (size_t)j] = r[2]; node[2 * (size_t)j + 1] = r[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) wide[4 * (size_t)j + i] = r[i];
}

int main() {
  for (uint32_t n : {1310720u, 16u * 1310720u}) {
    std::vector<uint32_t> h(n); std::iota(h.begin(), h.end(), 0u); std::mt19937 g(7); std::shuffle(h.begin(), h.end(), g);
    float4 *rec, *a, *b, *c, *leaf, *node, *wide; uint32_t* ord; float* sink;
    CK(hipMalloc(&rec, (size_t)n * 128)); CK(hipMalloc(&a, (size_t)n * 48)); CK(hipMalloc(&b, (size_t)n * 48)); CK(hipMalloc(&c, (size_t)n * 48));
    CK(hipMalloc(&leaf, (size_t)n * 32)); CK(hipMalloc(&node, (size_t)n * 32)); CK(hipMalloc(&wide, (size_t)n * 64)); CK(hipMalloc(&ord, (size_t)n * 4));
    CK(hipMalloc(&sink, (size_t)n * 4));
    CK(hipMemset(rec, 0, (size_t)n * 128)); CK(hipMemcpy(ord, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    for (int v = 0; v < 6; ++v) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&]() {
        dim3 gr((n + 255) / 256), bl(256);
        if (v == 0) hipLaunchKernelGGL(k<0>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
        if (v == 1) hipLaunchKernelGGL(k<1>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
        if (v == 2) hipLaunchKernelGGL(k<2>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
        if (v == 3) hipLaunchKernelGGL(k<3>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
        if (v == 4) hipLaunchKernelGGL(k<4>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
        if (v == 5) hipLaunchKernelGGL(k<5>, gr, bl, 0, 0, rec, ord, n, a, b, c, leaf, node, wide, sink);
      };
      for (int i = 0; i < 3; ++i) launch();
      CK(hipDeviceSynchronize());
      const int reps = 10;
      CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      const double rd = (v == 0 || v == 4) ? 0.0 : (v == 1 ? 128.0 : 132.0), wr = (v == 3) ? 0.0 : 272.0;
      printf("n %9u variant %d : %8.2f us  read %6.1f MB write %6.1f MB -> %6.2f TB/s\n", n, v, ms * 1e3, rd * n / 1e6, wr * n / 1e6,
             (rd + wr) * n / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(rec)); CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(leaf)); CK(hipFree(node)); CK(hipFree(wide)); CK(hipFree(ord)); CK(hipFree(sink));
  }
  return 0;
}
