// ubench_bounds.hip — why does k_bounds take 30 us for 100 MB at 1.31 M triangles whatever its gather looks like?
// De-indexed mesh (vidx = 0, 1, 2, ...), caches flushed between runs by a 1 GB fill.  Variants of the slot-wise kernel:
//   0: as k_bounds (idx -> vertex gather -> LDS -> centroid store), grid capped at 2048 blocks
//   1: grid not capped (5120 blocks, one step per wave)
//   2: no centroid stores             3: idx loads only (no vertex gather, no stores)
//   4: as 1 with two steps' loads in flight per wave (128 triangles per wave step)
//   5: pure stream of the same bytes: float4 loads of idx + pos, float4 stores of 12 B/triangle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ pos, const uint32_t* __restrict__ vidx, uint32_t n_tri,
                                         float* __restrict__ centroids, float* __restrict__ sink) {
  __shared__ float s_vtx[4][2][192 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  const size_t n_slots = 3 * (size_t)n_tri;
  if (V == 5) {
    const size_t n4 = (size_t)n_tri * 12 / 4;       // float4s of idx + pos together (48 B per triangle)
    const float4* a = reinterpret_cast<const float4*>(vidx); const float4* b = reinterpret_cast<const float4*>(pos);
    float4* c = reinterpret_cast<float4*>(centroids);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      const float4 x = (i < n4 / 4) ? a[i] : b[i - n4 / 4];
      if ((i & 3) == 0) c[i / 4] = x; else acc += x.x;
    }
    if (acc == 123.456f) sink[0] = acc;
    return;
  }
  constexpr int U = (V == 4) ? 2 : 1;
  for (size_t t0 = ((size_t)blockIdx.x * 4 + wave) * 64 * U; t0 < n_tri; t0 += (size_t)gridDim.x * 256 * U) {
    uint32_t ix[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < 3; ++r) { const size_t sl = 3 * (t0 + 64 * u) + 64 * r + lane; ix[u][r] = (sl < n_slots) ? vidx[sl] : 0u; }
    if (V == 3) { for (int u = 0; u < U; ++u) acc += (float)(ix[u][0] + ix[u][1] + ix[u][2]); continue; }
    float v[U][3][3];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < 3; ++r) { const float* p = pos + 3 * (size_t)ix[u][r]; v[u][r][0] = p[0]; v[u][r][1] = p[1]; v[u][r][2] = p[2]; }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < 3; ++r) { float* w = s_vtx[wave][u] + 3 * (64 * r + lane); w[0] = v[u][r][0]; w[1] = v[u][r][1]; w[2] = v[u][r][2]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t t = t0 + 64 * u + lane;
      if (t < n_tri) {
        const float* q = s_vtx[wave][u] + 9 * lane;
        const float cx = ((q[0] + q[3]) + q[6]) / 3.0f, cy = ((q[1] + q[4]) + q[7]) / 3.0f, cz = ((q[2] + q[5]) + q[8]) / 3.0f;
        if (V != 2) { centroids[3 * t + 0] = cx; centroids[3 * t + 1] = cy; centroids[3 * t + 2] = cz; }
        acc += cx + cy + cz;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 1310720u;      // 99614720 = cfg 5
  std::vector<uint32_t> hi(3 * (size_t)n); for (size_t i = 0; i < hi.size(); ++i) hi[i] = (uint32_t)i;
  std::vector<float> hp(9 * (size_t)n); for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)(i % 977) * 0.001f;
  float *pos, *cen, *sink; uint32_t* idx; char* flush;
  CK(hipMalloc(&pos, hp.size() * 4)); CK(hipMalloc(&idx, hi.size() * 4)); CK(hipMalloc(&cen, (size_t)n * 12)); CK(hipMalloc(&sink, 256));
  const size_t flush_bytes = 1ull << 30; CK(hipMalloc(&flush, flush_bytes));
  CK(hipMemcpy(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int cold = 1; cold >= 0; --cold)
    for (int v = 0; v < 6; ++v) {
      const int grid = (v == 1) ? (int)((n + 255) / 256) : (v == 4) ? (int)((n + 511) / 512) : (argc > 2 ? atoi(argv[2]) : 2048);
      float best = 1e9f, sum = 0.f; const int reps = 6;
      for (int r = 0; r < reps + 1; ++r) {
        if (cold) CK(hipMemsetAsync(flush, r, flush_bytes, 0));
        CK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        if (v == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        if (v == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, pos, idx, n, cen, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { best = ms < best ? ms : best; sum += ms; }
      }
      printf("%s variant %d grid %5d : mean %7.2f us  best %7.2f us\n", cold ? "cold" : "warm", v, grid, sum / reps * 1e3, best * 1e3);
    }
  return 0;
}
