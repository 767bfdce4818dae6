// ubench_gather.hip — what does a divergent 64-byte-record gather cost per CU on gfx950?
// Each lane chases a pseudo-random chain of 64 B records (like BVH nodes). Variants:
//   0: 4 x dwordx4 per lane from the lane's own record (current k_render pattern)
//   1: 1 x dwordx4 per lane (quarter of the record)
//   2: quad-cooperative: 4 lanes fetch one record with one dwordx4 each; 4 instructions cover the wave's 64 records
//   3: 2 x dwordx4 per lane (half record)
//   4: 3 x dwordx4 + 1 x dwordx2 per lane (the 56 useful bytes of a two-child record)
//   5: 3 x dwordx4 per lane (48 bytes)
//   6 / 7: as 0, but every 4th / 2nd step goes to the OTHER 64-byte half of the same 128-byte line (a parent and its near child laid out as a pair)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, uint32_t n_rec, int steps, float* out) {
  const int lane = threadIdx.x & 63;
  uint32_t cur = (blockIdx.x * 256 + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float4 a, b, c, d;
    if (V == 0 || V == 6 || V == 7) { const float4* p = tab + 4 * (size_t)cur; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
    else if (V == 1) { const float4* p = tab + 4 * (size_t)cur; a = p[0]; b = c = d = a; }
    else if (V == 3) { const float4* p = tab + 4 * (size_t)cur; a = p[0]; b = p[1]; c = a; d = b; }
    else if (V == 4) { const float4* p = tab + 4 * (size_t)cur; a = p[0]; b = p[1]; c = p[2]; const float2 e = *reinterpret_cast<const float2*>(p + 3); d = make_float4(e.x, e.y, e.x, e.y); }
    else if (V == 5) { const float4* p = tab + 4 * (size_t)cur; a = p[0]; b = p[1]; c = p[2]; d = a; }
    else {
      float4 r[4];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const uint32_t want = (uint32_t)__shfl((int)cur, k4 * 16 + (lane >> 2));
        r[k4] = tab[4 * (size_t)want + (lane & 3)];
      }
      // (a real kernel would transpose back; here just keep everything live)
      a = r[0]; b = r[1]; c = r[2]; d = r[3];
    }
    acc += a.x + b.y + c.z + d.w;
    const uint32_t nxt = (__float_as_uint(a.w) ^ (uint32_t)s * 0x9E3779B9u ^ cur * 747796405u);
    if (V == 6 && ((nxt >> 20) & 3u) == 0u) cur = cur ^ 1u;
    else if (V == 7 && ((nxt >> 20) & 1u) == 0u) cur = cur ^ 1u;
    else cur = nxt % n_rec;   // data-dependent next record
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  int ncu = 256;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
  const int steps = 400;
  for (size_t mb : {1ul, 16ul, 84ul, 512ul}) {
    if (getenv("UBENCH_MB") && (size_t)atoi(getenv("UBENCH_MB")) != mb) continue;
    const uint32_t n_rec = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<float4> h((size_t)n_rec * 4);
    for (size_t i = 0; i < h.size(); ++i) { uint32_t x = (uint32_t)i * 2246822519u + 12345u; x ^= x >> 15; x *= 2654435761u; h[i] = make_float4(1e-9f, 1e-9f, 1e-9f, __builtin_bit_cast(float, x & 0x3fffffffu)); }
    float4* d; float* o; CK(hipMalloc(&d, h.size() * sizeof(float4))); CK(hipMalloc(&o, (size_t)ncu * 8 * 256 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    for (int bpc : {3}) {
      for (int v : {0, 6, 7}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
          dim3 g(ncu * bpc), b(256);
          if (v == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, d, n_rec, steps, o);
          if (v == 7) hipLaunchKernelGGL(k<7>, g, b, 0, 0, d, n_rec, steps, o);
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_steps_per_cu = (double)bpc * 4 * steps;
        printf("table %4zu MB  blocks/CU %d  variant %d : %8.3f ms  -> %7.1f ns per wave-step per CU, %6.2f us per dependent step\n", mb, bpc, v, ms,
               ms * 1e6 / wave_steps_per_cu, ms * 1e3 / steps);
      }
    }
    CK(hipFree(d)); CK(hipFree(o));
  }
  return 0;
}
