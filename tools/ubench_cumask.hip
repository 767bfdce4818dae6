// ubench_cumask.hip - does hipExtStreamCreateWithCUMask confine a launch to the CUs of its mask on this box?  Prints the return
// codes and, per launch, the set of (XCC, SE, CU) ids its blocks ran on.
// Observed (ROCm 7.2.0 image, MI355X, round 3): the process dies with a segmentation fault INSIDE hipExtStreamCreateWithCUMask,
// before anything is launched (profiles/r3/ab_express_lanes.txt) - so nothing in the library uses CU masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void k(uint32_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xFFFFu) | ((xcc & 0xFu) << 16);
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int n_cu = pr.multiProcessorCount;
  printf("CUs %d\n", n_cu);
  std::vector<uint32_t> mx((n_cu + 31) / 32, 0u), mb((n_cu + 31) / 32, 0u);
  for (int cu = 0; cu < n_cu; ++cu) ((cu % 8 == 0) ? mx : mb)[cu / 32] |= 1u << (cu % 32);
  hipStream_t xs = nullptr, bs = nullptr;
  printf("creating the express stream (%zu mask words)\n", mx.size());
  hipError_t e1 = hipExtStreamCreateWithCUMask(&xs, (uint32_t)mx.size(), mx.data());
  printf("-> %s\n", hipGetErrorString(e1));
  hipError_t e2 = hipExtStreamCreateWithCUMask(&bs, (uint32_t)mb.size(), mb.data());
  printf("create: %s / %s   last error: %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(hipGetLastError()));
  uint32_t* d; hipMalloc(&d, 4096 * 4);
  std::vector<uint32_t> h(4096);
  hipStream_t plain; hipStreamCreate(&plain);
  hipStream_t ss[3] = {plain, xs, bs}; const char* names[3] = {"plain", "express mask (every 8th CU)", "bulk mask (the others)"};
  for (int i = 0; i < 3; ++i) {
    if (!ss[i]) continue;
    hipMemset(d, 0xFF, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, ss[i], d, 2000);
    hipError_t el = hipGetLastError();
    hipError_t es = hipStreamSynchronize(ss[i]);
    hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus;   // HW_ID: [11:8] cu id, [14:13] se id (gfx9 layout), + xcc
    for (int b = 0; b < 2048; ++b) cus.insert(h[b] & 0xF7F00u);
    printf("%-30s launch %s sync %s  distinct (xcc, se, sh, cu) %zu\n", names[i], hipGetErrorString(el), hipGetErrorString(es), cus.size());
  }
  return 0;
}
