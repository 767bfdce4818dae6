// pmc_calib.hip — known byte counts for the PMC counters the bench's roofline uses (run under scripts/pmc_passes.sh):
//   calib_stream : reads 1 GiB once, 16 B per lane, fully coalesced            -> bytes per TCC_REQ / TCC_MISS / FETCH_SIZE unit
//   calib_gather : 4 Mi lanes fetch one random 64-byte record each (4 x 16 B) from an 80 MB table -> counts per divergent record
// Each kernel runs 3 times; pmc_summary.py prints the mean per dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ src, size_t n16, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather(const float4* __restrict__ tab, uint32_t n_rec, float* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = t * 2654435761u + 12345u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  const float4* p = tab + 4 * (size_t)(x % n_rec);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3];
  const float acc = a.x + b.y + c.z + d.w;
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t n16 = (size_t)1 << 26;                       // 1 GiB of float4
  float4* src; float* out;
  CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, n16 * 16));
  const uint32_t n_rec = 1310720u;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream, dim3(256 * 8), dim3(256), 0, 0, src, n16, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_gather, dim3((4u << 20) / 256), dim3(256), 0, 0, src, n_rec, out);
    CK(hipDeviceSynchronize());
  }
  printf("calib_stream: %zu bytes per dispatch; calib_gather: %u records of 64 B per dispatch (table %u records)\n", n16 * 16, 4u << 20, n_rec);
  return 0;
}
