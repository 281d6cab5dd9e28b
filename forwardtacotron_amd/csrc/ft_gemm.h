// Internal descriptors for the exact-f32 MFMA GEMM / shifted-row (conv) kernels.
#pragma once
#include "ft_common.h"

// logical row r = (b, t), b = r / Tlog, t = r % Tlog:
//   physical row = b*bstride + (t+shift)*tstride, valid iff 0 <= t+shift < Tvalid ; invalid rows read as zeros.
//   shift = shift0 + tap*shift_step.
// batch-major [B,Tbuf,C]: bstride = Tbuf, tstride = 1 ; time-major [T,B,C]: bstride = 1, tstride = B.
struct FtRowMap {
  int Tlog, bstride, tstride, Tvalid, shift0, shift_step;
};

static inline FtRowMap ft_rowmap_identity(int rows) {
  int n = rows > 0 ? rows : 1;
  FtRowMap m = {n, 0, 1, n, 0, 0};
  return m;
}
// rows = B*T logical (b,t) positions stored time-major ([T,B,C]) when tm_B > 0, else plain row-major
static inline FtRowMap ft_rowmap_layout(int rows, int tm_B) {
  if (tm_B <= 0) return ft_rowmap_identity(rows);
  int T = rows / tm_B;
  FtRowMap m = {T > 0 ? T : 1, 1, tm_B, T, 0, 0};
  return m;
}

// C[cmap(M),N] (+)= sum_{tap} Amap_tap(A)[M,K] * B_tap   (+bias, relu, affine)
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
  FtRowMap cmap;        // output row permutation (Tlog/bstride/tstride only); identity by default
  int relu, accumulate, a_vec, b_vec;
  // NN form only: every row of A is readable up to the next multiple of 4 of K (values there are ignored), so a K that
  // is not a multiple of 4 can still take the 16-B-load path
  int a_rowpad;
  // strided batch (attention: one GEMM per (batch item, head)); only for single-task launches.
  // instance z = blockIdx.z -> (z0, z1) = (z / nz1, z % nz1); X += z0*sX0 + z1*sX1 (floats)
  int nz, nz1;
  long sA0, sA1, sB0, sB1, sC0, sC1;
  // BatchNorm statistics from the epilogue (128x128 bf16-split kernel only; see ft_launch_gemm_rows): per M-tile and
  // column, (sum, sum of squares) of the stored values (after bias / ReLU) over the tile's rows with t = row % amap.Tlog
  // < stat_tvalid, as doubles at stat[((long)mtile * stat_ld + stat_col0 + col) * 2 + {0,1}] -- the layout of the
  // chunked partials ft_bn.hip finalizes.  stat = null: off.
  double* stat;
  int stat_ld, stat_col0, stat_tvalid;
};
#define FT_MAX_TASKS 16
struct FtGemmBatch {
  FtGemmTask t[FT_MAX_TASKS];
  // chain > 1: tasks 0..chain-1 are ONE product, accumulated in registers: C = sum_i sum_tap shift(A_i) * B_i,tap with
  // task 0's M, N, output and epilogue (conv-bank data gradient: every member adds into the same dx)
  int chain;
  int stat_fused;       // out: 1 if the launched kernel produced the tasks' BatchNorm statistics
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
  // rows of A / B are readable up to the next multiple of 4 of M / N (those columns only reach masked outputs)
  int rowpad;
  // strided batch (see FtGemmTask); dst += z0*sD0 + z1*sD1
  int nz, nz1;
  long sA0, sA1, sB0, sB1, sD0, sD1;
  // conv-bank mode (bankC > 0): A's M columns are the K = M / bankC members of a CBHG conv bank side by side, member
  // kk = m / bankC + 1 has kernel size kk: tap j < kk only, B row shift j - kk/2, A rows t < (kk odd ? bankTodd :
  // amap.Tvalid); the result goes to bank_dst[kk-1] in torch layout [bankC][N][kk].  bankC % tile == 0.
  int bankC, bankTodd;
  float* bank_dst[FT_MAX_TASKS];
};

// Workgroup -> (M-tile, tap, split) of a TN launch.  Plain tasks: blockIdx.x = M-tile, blockIdx.z = (instance*taps +
// tap)*S + split.  Conv-bank mode: member kk only has taps j < kk, so the grid enumerates the LIVE (member, tap, tile)
// triples along x (kk ascending, then tap, then the member's M-tiles) and blockIdx.z = split -- no empty workgroups,
// equal work per workgroup (the round-robin XCD assignment stays balanced), and neighbours share their dy tile.
struct FtTnWho {
  int mtile, zts, s, kk;      // zts = instance*taps + tap ; kk = member (bank mode) or 0
};
__device__ __forceinline__ FtTnWho ft_tn_who(const FtGemmTNTask& T, int S, int BM) {
  FtTnWho w;
  if (T.bankC > 0) {
    const int tpm = T.bankC / BM;                 // M-tiles per member
    const int q = blockIdx.x / tpm, sub = blockIdx.x - q * tpm;
    int kk = 1, first = 0;                        // first = kk(kk-1)/2 = index of (kk, tap 0)
    while (first + kk <= q) {
      first += kk;
      ++kk;
    }
    w.kk = kk;
    w.zts = q - first;
    w.mtile = (kk - 1) * tpm + sub;
    w.s = blockIdx.z;
  } else {
    w.kk = 0;
    w.mtile = blockIdx.x;
    w.zts = blockIdx.z / S;
    w.s = blockIdx.z - w.zts * S;
  }
  return w;
}

int ft_launch_gemm_rows(FtGemmBatch* batch, int ntasks, bool b_ncontig, hipStream_t stream);
// fp32 on the bf16 matrix pipe (exact 3-way operand split, ft_gemm_b3.hip); NT + FAST launches, FT_GEMM_B3=0 disables
int ft_launch_gemm_rows_b3(const FtGemmBatch& batch, bool big, dim3 grid, hipStream_t stream);
int ft_launch_gemm_tn_b3(const FtGemmTNTask& t, float* slab, int S, int rows_per_split, int tm, dim3 grid,
                         hipStream_t stream);
// ft_planes.hip: planes pointer for a B operand (first row / k chunk of the launch) or nullptr
const void* ft_planes_lookup(const float* b, long ldb, long rows_needed);
// ft_capi_core.hip: workgroup slots (2 per usable CU) of a stream -- 512 unless it is CU-limited
void ft_note_stream_slots(hipStream_t s, int slots);
int ft_stream_slots(hipStream_t s);
bool ft_gemm_b3_enabled();
// 0 = fp32-exact (default), 1 = bf16 operands / fp32 accumulate on every NT-form fast launch (ft_set_gemm_precision)
int ft_gemm_precision();
int ft_launch_gemm_tn(const FtGemmTNTask& task, float* workspace, size_t workspace_floats,
                      hipStream_t stream);
size_t ft_gemm_tn_workspace_floats(const FtGemmTNTask& task);
// dst[m*ldm + n] = sum_{s<S} slab[s][m][n]  (fixed order)
int ft_launch_slab_sum(const float* slab, float* dst, int M, int N, int S, long ldm, hipStream_t stream);
