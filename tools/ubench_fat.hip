// ubench_fat.hip — what would a two-levels-per-fetch node record cost?  (round 3)
// Every lane chases a pseudo-random chain through N records of STRIDE bytes and reads the first C 16-byte chunks of each
// (C independent dwordx4 loads issued together, one dependent round trip per record).  N = 1 310 720 records, like the
// internal nodes of cfg 3's tree: STRIDE 64 = today's `wide` table (84 MB), 128 / 192 / 256 = records that also hold
// grandchildren (168 / 252 / 336 MB).  Compared per TWO tree levels: 2 x (STRIDE 64, C 4) against 1 x (STRIDE 256, C 11).
//   throughput mode: 3 blocks of 256 threads per CU (k_render's occupancy);  latency mode: ONE wave per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int C>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, uint32_t n_rec, uint32_t stride16, int steps, uint32_t lane_mask_mod, float* out) {
  uint32_t cur = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.f;
  if (lane_mask_mod > 1 && (threadIdx.x % lane_mask_mod) != 0) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f; return; }   // only every k-th lane active
  for (int s = 0; s < steps; ++s) {
    const float4* p = tab + (size_t)stride16 * cur;
    float4 r[C];
#pragma unroll
    for (int i = 0; i < C; ++i) r[i] = p[i];
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) { acc += r[i].x + r[i].y; h ^= __float_as_uint(r[i].w) + (uint32_t)i; }
    cur = (h ^ (uint32_t)s * 0x9E3779B9u ^ cur * 747796405u) % n_rec;     // data-dependent next record: needs EVERY chunk
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int C>
float run(const float4* d, uint32_t n_rec, uint32_t stride16, int steps, int blocks, int threads, uint32_t lmm, float* o) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), 0, 0, d, n_rec, stride16, steps, lmm, o); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), 0, 0, d, n_rec, stride16, steps, lmm, o);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const uint32_t n_rec = 1310720u;
  const int steps = 300;
  float* o; CK(hipMalloc(&o, (size_t)ncu * 8 * 256 * 4));
  for (uint32_t stride : {64u, 128u, 192u, 256u}) {
    const size_t n16 = (size_t)n_rec * (stride / 16);
    std::vector<float4> h(n16);
    for (size_t i = 0; i < n16; ++i) { uint32_t x = (uint32_t)i * 2246822519u + 12345u; x ^= x >> 15; x *= 2654435761u; h[i] = make_float4(1e-9f, 1e-9f, 1e-9f, __builtin_bit_cast(float, x & 0x3fffffffu)); }
    float4* d; CK(hipMalloc(&d, n16 * sizeof(float4)));
    CK(hipMemcpy(d, h.data(), n16 * sizeof(float4), hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; ++mode) {   // 0: 3 blocks x 256 per CU, all lanes; 1: the same, every 2nd lane active; 2: one wave per CU
      const int blocks = mode == 2 ? ncu : ncu * 3, threads = mode == 2 ? 64 : 256;
      const uint32_t lmm = mode == 1 ? 2u : 1u;
      for (int c : {3, 4, 7, 8, 11, 12, 16}) {
        if ((uint32_t)c * 16u > stride) continue;
        float ms = 0.f;
        switch (c) {
          case 3: ms = run<3>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 4: ms = run<4>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 7: ms = run<7>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 8: ms = run<8>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 11: ms = run<11>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 12: ms = run<12>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
          case 16: ms = run<16>(d, n_rec, stride / 16, steps, blocks, threads, lmm, o); break;
        }
        const double waves_per_cu = mode == 2 ? 1.0 : 12.0;
        printf("stride %3u B (%3zu MB)  chunks %2d  %-28s: %8.3f ms -> %7.1f ns per wave-step per CU, %6.3f us per dependent step\n", stride,
               n16 * 16 >> 20, c, mode == 0 ? "12 waves/CU, 64 lanes" : mode == 1 ? "12 waves/CU, 32 lanes active" : "1 wave/CU (latency)",
               ms, ms * 1e6 / (waves_per_cu * steps), ms * 1e3 / steps);
      }
    }
    CK(hipFree(d));
  }
  return 0;
}
