// GEMM lab: C[M,N] = A[M,K] * B[N,K]^T, exact f32 MFMA, same tiling as ft_gemm_rows_kernel<2,2,NT,FAST> -- used to
// find what keeps the product kernel at ~60 % of the MFMA rate.  hipcc -O3 --offload-arch=gfx950 gemm_lab.hip -o gemm_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

// VAR: 0 baseline, 1 no global loads in the loop, 2 also no LDS stores, 3 also no LDS reads (MFMA only)
template <int VAR>
__global__ __launch_bounds__(256) void k_base(const float* __restrict__ A, const float* __restrict__ B,
                                              float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, LDA = BM + 1, LDB = BN + 1, PA = 4, PB = 4;
  constexpr int STAGE = BK * LDA + BK * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, kq = tid & 7, rr = tid >> 3;
  const int nch = K / BK;
  float4 ra[PA], rb[PB];
  auto load_stage = [&](int c) {
    const int k = c * BK + 4 * kq;
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const float4*>(A + (long)(m0 + rr + 32 * p) * K + k);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const float4*>(B + (long)(n0 + rr + 32 * p) * K + k);
  };
  auto store_stage = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int m = rr + 32 * p;
      As[(4 * kq + 0) * LDA + m] = ra[p].x;
      As[(4 * kq + 1) * LDA + m] = ra[p].y;
      As[(4 * kq + 2) * LDA + m] = ra[p].z;
      As[(4 * kq + 3) * LDA + m] = ra[p].w;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      int n = rr + 32 * p;
      Bs[(4 * kq + 0) * LDB + n] = rb[p].x;
      Bs[(4 * kq + 1) * LDB + n] = rb[p].y;
      Bs[(4 * kq + 2) * LDB + n] = rb[p].z;
      Bs[(4 * kq + 3) * LDB + n] = rb[p].w;
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  load_stage(0);
  store_stage(0);
  if (nch > 1) load_stage(1);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int cur = c & 1;
    if (c + 1 < nch) {
      if (VAR < 2) store_stage(cur ^ 1);
      if (VAR < 1 && c + 2 < nch) load_stage(c + 2);
    }
    const float* ap = smem + cur * STAGE + half * LDA + wm * 64 + l31;
    const float* bp = smem + cur * STAGE + BK * LDA + half * LDB + wn * 64 + l31;
    float a0[TM], b0[TN], a1[TM], b1[TN];
    if (VAR < 3) {
      for (int i = 0; i < TM; ++i) a0[i] = ap[32 * i];
      for (int j = 0; j < TN; ++j) b0[j] = bp[32 * j];
    } else {
      for (int i = 0; i < TM; ++i) a0[i] = a1[i] = ra[i].x;
      for (int j = 0; j < TN; ++j) b0[j] = b1[j] = rb[j].y;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 2; ks += 2) {
      if (VAR < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = ap[2 * (ks + 1) * LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = bp[2 * (ks + 1) * LDB + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR < 3 && ks + 2 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = ap[2 * (ks + 2) * LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = bp[2 * (ks + 2) * LDB + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

template <typename F>
double tf(F launch, double flops, int reps = 5) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return flops * reps / (ms * 1e-3) / 1e12;
}

#ifdef LAB_V2
#include "gemm_lab_v2.h"
#endif
#ifdef LAB_B3
#include <math.h>
#include "gemm_lab_b3.h"
#endif

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 26880, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
  float *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * K * 4);
  (void)hipMalloc(&B, (size_t)N * K * 4);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = (float)((i * 37 % 101) - 50) / 64.f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 53 % 89) - 44) / 64.f;
  (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K;
  dim3 grid(M / 128, N / 128);
  printf("M=%d N=%d K=%d\n", M, N, K);
  printf("base            : %.1f TF\n", tf([&] { hipLaunchKernelGGL(k_base<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("no global loads : %.1f TF\n", tf([&] { hipLaunchKernelGGL(k_base<1>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("no LDS stores   : %.1f TF\n", tf([&] { hipLaunchKernelGGL(k_base<2>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("MFMA only       : %.1f TF\n", tf([&] { hipLaunchKernelGGL(k_base<3>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
#ifdef LAB_V2
  lab_v2(A, B, C, M, N, K, fl, ha, hb);
#endif
#ifdef LAB_B3
  lab_b3(A, B, C, M, N, K, fl);
#endif
  return 0;
}
