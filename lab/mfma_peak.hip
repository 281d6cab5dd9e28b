// Pure-MFMA throughput probe: what does v_mfma_f32_32x32x2_f32 (and 16x16x4) sustain on this part, with
// 1 / 2 waves per SIMD, no memory traffic?   hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double run(F launch, double flop_per_launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return flop_per_launch * 5 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  const int iters = 4000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
    double f32 = 2.0 * 32 * 32 * 2, f16 = 2.0 * 16 * 16 * 4;
    printf("%d wave(s)/SIMD:\n", wgs_per_cu);
    printf("  32x32x2 f32, 4 acc: %.1f TF\n",
           run([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
               f32 * 8 * 4 * iters * 4.0 * grid));
    printf("  32x32x2 f32, 2 acc: %.1f TF\n",
           run([&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
               f32 * 8 * 2 * iters * 4.0 * grid));
    printf("  32x32x2 f32, 1 acc: %.1f TF\n",
           run([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
               f32 * 8 * 1 * iters * 4.0 * grid));
    printf("  16x16x4 f32, 4 acc: %.1f TF\n",
           run([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
               f16 * 8 * 4 * iters * 4.0 * grid));
    printf("  16x16x4 f32, 1 acc: %.1f TF\n",
           run([&] { hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
               f16 * 8 * 1 * iters * 4.0 * grid));
  }
  return 0;
}
