// Internal descriptors for the exact-f32 MFMA GEMM / shifted-row (conv) kernels.
#pragma once
#include "ft_common.h"

// logical row r = (b, t), t in [0,Tlog):  physical row = b*Tstride + (t+shift),
// valid iff 0 <= t+shift < Tvalid ; invalid rows read as zeros.  shift = shift0 + tap*shift_step.
struct FtRowMap {
  int Tlog, Tstride, Tvalid, shift0, shift_step;
};

static inline FtRowMap ft_rowmap_identity(int rows) {
  FtRowMap m = {rows > 0 ? rows : 1, rows > 0 ? rows : 1, rows > 0 ? rows : 1, 0, 0};
  return m;
}

// C[M,N] (+)= sum_{tap} Amap_tap(A)[M,K] * B_tap   (+bias, relu)
//   B_NCONTIG = false: B_tap[n][k] at B + tap*b_tap_stride + n*ldb + k   ("NT")
//   B_NCONTIG = true : B_tap[k][n] at B + tap*b_tap_stride + k*ldb + n   ("NN")
struct FtGemmTask {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* scale;   // optional per-column affine applied AFTER relu (eval-mode BatchNorm fold)
  const float* shift;
  long lda, ldb, ldc, b_tap_stride;
  int M, N, K, taps;
  FtRowMap amap;
  int relu, accumulate, a_vec, b_vec;
};
#define FT_MAX_TASKS 16
struct FtGemmBatch {
  FtGemmTask t[FT_MAX_TASKS];
};

// out_tap[m][n] = sum_r Amap(A)[r][m] * Bmap_tap(B)[r][n],  r over R logical rows   ("TN", split over rows)
// result written to dst[m*ldm + n*ldn + tap*ldj]  (deterministic slab + reduce)
struct FtGemmTNTask {
  const float* A;
  const float* B;
  float* dst;
  long lda, ldb;
  long ldm, ldn, ldj;
  int M, N, R, taps;
  FtRowMap amap, bmap;
  int a_vec, b_vec, accumulate;
};

int ft_launch_gemm_rows(FtGemmBatch* batch, int ntasks, bool b_ncontig, hipStream_t stream);
int ft_launch_gemm_tn(const FtGemmTNTask& task, float* workspace, size_t workspace_floats,
                      hipStream_t stream);
size_t ft_gemm_tn_workspace_floats(const FtGemmTNTask& task);
