// Which CUs does a CU-masked stream reach?  Each workgroup records the XCC / SE / CU id of the CU it ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
__global__ void who(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
  const int NB = 4096;
  unsigned* d; hipMalloc(&d, NB * 2 * sizeof(unsigned));
  std::vector<unsigned> h(NB * 2);
  const uint32_t masks[][8] = {
      {0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu, 0x0FFFFFFFu},
      {0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u, 0xFFFFFFF0u},
      {0x0FFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu},
      {0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu},
      {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u},
      {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0}};
  for (int m = -1; m < 6; ++m) {
    hipStream_t s;
    if (m < 0) hipStreamCreate(&s);
    else if (hipExtStreamCreateWithCUMask(&s, 8, masks[m]) != hipSuccess) { printf("mask %d: create failed\n", m); continue; }
    hipLaunchKernelGGL(who, dim3(NB), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, NB * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int perx[8] = {0};
    std::set<unsigned> perxcu[8];
    for (int i = 0; i < NB; ++i) {
      unsigned x = h[2 * i] & 0xF, hw = h[2 * i + 1];
      unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned id = (x << 12) | (se << 8) | (sh << 4) | cu;
      cus.insert(id); perxcu[x & 7].insert(id);
    }
    printf("mask %d: %zu distinct CUs; per XCC:", m, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", perxcu[x].size());
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
