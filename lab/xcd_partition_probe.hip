// Does a CU mask that selects whole XCDs partition the chip?  (1) where do the workgroups of a launch on a stream masked to
// XCDs 0-3 go (workgroup i -> which XCC id);  (2) while a kernel that fills XCDs 0-3 (one 100-KB-LDS workgroup per CU) spins
// for ~3 ms on that stream, how long does a 4096-workgroup kernel take on (a) an unmasked stream, (b) a stream masked to
// XCDs 4-7.   hipcc -O3 --offload-arch=gfx950 lab/xcd_partition_probe.hip -o lab/xcd_partition_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ unsigned xcc() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ void where_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc(); }

__global__ __launch_bounds__(512) void hog_kernel(unsigned* out, long long cycles) {
  __shared__ float big[40000];                       // 160 KB: the CU takes nothing else
  big[threadIdx.x] = 1.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) out[blockIdx.x] = xcc() + (unsigned)big[1];
}
// the aligned layout of the persistent recurrences: 256 workgroups, those that land on XCC >= 4 leave at once, the others
// (32 per XCD, one per CU) stay
__global__ __launch_bounds__(512) void hog4_kernel(unsigned* out, long long cycles) {
  __shared__ float big[40000];
  if (xcc() >= 4) return;
  big[threadIdx.x] = 1.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) out[blockIdx.x] = xcc() + (unsigned)big[1];
}

__global__ void small_kernel(float* buf) {
  __shared__ float s[8192];                          // 32 KB
  s[threadIdx.x] = (float)blockIdx.x;
  __syncthreads();
  float a = s[(threadIdx.x * 7) & 255];
  for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f;
  buf[blockIdx.x * 256 + threadIdx.x] = a;
}

int g_mapping = 0;      // 0: bit i = CU i/8 of XCD i%8 ; 1: bit i = CU i%32 of XCD i/32
hipStream_t masked(unsigned xcd_bits) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 256; ++i) if ((xcd_bits >> (g_mapping ? (i >> 5) : (i & 7))) & 1) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) return nullptr;
  return s;
}

int main(int argc, char** argv) {
  g_mapping = argc > 1 ? atoi(argv[1]) : 0;
  printf("mask mapping %d\n", g_mapping);
  unsigned* d; float* buf;
  CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&buf, 4096 * 256 * 4));
  hipStream_t lo = masked(0x0F), hi = masked(0xF0), plain;
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  if (!lo || !hi) { printf("mask stream creation failed\n"); return 1; }
  std::vector<unsigned> h(4096);
  for (int which = 0; which < 3; ++which) {
    hipStream_t st = which == 0 ? plain : (which == 1 ? lo : hi);
    CK(hipMemset(d, 0xff, 4096 * 4));
    hipLaunchKernelGGL(where_kernel, dim3(256), dim3(512), 0, st, d);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
    printf("%s: XCC of workgroups 0..31:", which == 0 ? "plain     " : (which == 1 ? "XCDs 0-3  " : "XCDs 4-7  "));
    for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
    int hist[16] = {0}; for (int i = 0; i < 256; ++i) hist[h[i] & 15]++;
    printf("  | per XCC:"); for (int i = 0; i < 8; ++i) printf(" %d", hist[i]); printf("\n");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long cyc = 300000;                    // wall_clock64 runs at 100 MHz: 3 ms
  for (int hogmask = 0; hogmask < 2; ++hogmask)
    for (int which = 0; which < 3; ++which) {
      hipStream_t hs = hogmask ? lo : plain;
      hipStream_t st = which == 0 ? plain : (which == 1 ? hi : lo);
      if (hs == st) continue;
      CK(hipDeviceSynchronize());
      // the hog: 128 workgroups, one per CU.  On the plain stream it lands on all 8 XCDs (16 CUs each); on `lo` it fills XCDs 0-3
      hipLaunchKernelGGL(hog_kernel, dim3(128), dim3(512), 0, hs, d, cyc);
      // give it time to become resident
      hipLaunchKernelGGL(where_kernel, dim3(1), dim3(64), 0, st, d + 2048);
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, st, buf);
      CK(hipEventRecord(e1, st));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("hog on %s, 4096 small workgroups on %s: %.3f ms\n", hogmask ? "XCDs 0-3 (fills them)" : "plain stream (half of every XCD)",
             which == 0 ? "plain" : (which == 1 ? "XCDs 4-7" : "XCDs 0-3"), ms);
    }
  for (int which = 0; which < 3; ++which) {
    hipStream_t st = which == 0 ? plain : (which == 1 ? hi : lo);
    hipStream_t hs = which == 0 ? lo : plain;
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(hog4_kernel, dim3(256), dim3(512), 0, hs, d, cyc);
    hipLaunchKernelGGL(where_kernel, dim3(1), dim3(64), 0, st, d + 2048);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, st, buf);
    CK(hipEventRecord(e1, st));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hog FILLS XCC 0-3 (aligned-layout style, launched unmasked), 4096 small workgroups on %s: %.3f ms\n",
           which == 0 ? "plain" : (which == 1 ? "mask hi" : "mask lo"), ms);
  }
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, plain, buf);
  CK(hipEventRecord(e0, plain));
  hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, plain, buf);
  CK(hipEventRecord(e1, plain));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("4096 small workgroups alone, plain: %.3f ms\n", ms);
  hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, hi, buf);
  CK(hipEventRecord(e0, hi));
  hipLaunchKernelGGL(small_kernel, dim3(4096), dim3(256), 0, hi, buf);
  CK(hipEventRecord(e1, hi));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("4096 small workgroups alone, XCDs 4-7: %.3f ms\n", ms);
  return 0;
}
